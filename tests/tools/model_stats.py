"""Lab tool: event counts of the token loop (windows, segments and why they end, bytes per segment, checked steps) from the
HOST model of the kernel source built with -DPZG_STATS, over a workload of bench.py.  Usage:
    python tests/tools/model_stats.py [l6_32k|l6_2k|l6_4k|l6_8k|fixed_4k|html|skewed_bytes|hetero_bin] [count]"""
import ctypes as C
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402

NAMES = ["windows (hot loop)", "tokens queued by clean windows", "segments", "segments in the fast body", "ended by the 128-byte limit",
         "ended by a source inside the segment", "ended by the queue running out", "sum of qn at segment start", "bytes of segments",
         "tokens of segments", "segments with a second pass", "head token for copy_match / bail", "checked steps",
         "strip spans", "records of strip spans", "phase-B rounds", "phase-A steps", "phase-B steps", "spans cut at a wrong start", "lanes that counted",
         "groups", "sequences of groups", "bytes of groups", "copy rounds", "solo sequences + long matches in groups", "groups with far matches", "dword steps", "byte steps", "sum of exact dependency depths",
         "spans ended by a lane out of steps", "spans laid out by the profile", "profile layouts the run-ups rejected"]


def build():
    d = os.path.join(ROOT, "tests", "model")
    so = os.path.join(d, "libpzgmodel_stats.so")
    extra = os.path.join(ROOT, "build", "model_stats_glue.cpp")
    os.makedirs(os.path.dirname(extra), exist_ok=True)
    with open(extra, "w") as f:
        f.write('#include "%s"\nPzgStats pzg_stats;\nextern "C" unsigned long long *pzm_stats(void) { return pzg_stats.v; }\n' % os.path.join(d, "model_harness.cpp"))
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DPZG_STATS", *os.environ.get("PZG_MODEL_FLAGS", "").split(), "-o", so, extra])
    return C.CDLL(so)


class R(C.Structure):
    _fields_ = [("status", C.c_int32), ("detail0", C.c_uint32), ("detail1", C.c_uint32), ("adler", C.c_uint32),
                ("out_len", C.c_uint64), ("in_used", C.c_uint64)]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "l6_32k"
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    M = build()
    M.pzm_stats.restype = C.POINTER(C.c_ulonglong)
    M.pzm_decompress.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(R)]
    tot_out = tot_in = 0
    for seed in range(cnt):
        if wl == "fixed_4k":
            t = corpus.zipf_text(4096, seed)
            co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
            z = co.compress(t) + co.flush()
        elif wl == "html":
            t = corpus.html_slice(32768, seed)
            z = zlib.compress(t, 6)
        elif wl == "hetero_bin":  # the binary-looking kind of bench.py's hetero workload (16-byte records)
            sys.path.insert(0, ROOT)
            import bench
            t, z = bench.hetero_blob(4 * seed + 3)
        elif wl in ("l6_2k", "l6_4k", "l6_8k"):
            nb = {"l6_2k": 2048, "l6_4k": 4096, "l6_8k": 8192}[wl]
            t = corpus.zipf_text(nb, seed)
            z = zlib.compress(t, 6)
        elif wl == "skewed_bytes":
            t = corpus.skewed_bytes(32768, seed)
            z = zlib.compress(t, 6)
        else:
            t = corpus.zipf_text(32768, seed)
            z = zlib.compress(t, 6)
        out = C.create_string_buffer(len(t) + 16)  # (l6_2k / l6_4k / l6_8k: small level-6 text streams)
        r = R()
        assert M.pzm_decompress(z, len(z), out, len(t), 11, C.byref(r)) == 0 and r.status == 0 and out.raw[:len(t)] == t
        tot_out += len(t)
        tot_in += len(z)
    v = M.pzm_stats()
    print(f"{wl}: {cnt} streams, {tot_in} B in, {tot_out} B out, per stream:")
    for i, n in enumerate(NAMES):
        print(f"  {n:42s} {v[i] / cnt:10.1f}")
    print(f"  bits per window {8 * tot_in / max(v[0], 1):.1f}; tokens per window {v[1] / max(v[0], 1):.2f}; bytes per segment {v[8] / max(v[2] - v[11], 1):.1f}; "
          f"tokens per segment {v[9] / max(v[2] - v[11], 1):.1f}; qn at segment start {v[7] / max(v[2], 1):.1f}")
    g = max(v[20], 1)
    print(f"  per group: sequences {v[21] / g:.1f}; bytes {v[22] / g:.1f}; copy rounds {v[23] / g:.2f}; dword steps {v[26] / g:.1f}; byte steps {v[27] / g:.2f}; "
          f"with far matches {v[25] / g:.2f}; solo per group {v[24] / g:.3f}; exact dependency depth {v[28] / g:.2f}")


if __name__ == "__main__":
    main()
