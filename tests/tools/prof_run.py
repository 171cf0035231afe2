"""Diagnostic: run a small batch through build/prof/libpzg.so (PZG_PROFILE build) and print the
per-phase s_memtime cycle breakdown.  Not a test, not shipped."""
import sys, os, zlib, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
from pure_zlib_amd import _ffi
_ffi.LIB_PATH = os.environ.get("PZG_PROF_LIB", os.path.join(ROOT, "build", "prof", "libpzg.so"))
import pure_zlib_amd as P
ctx = P.Context(0)
L = _ffi.lib()
L.pzg_prof_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
datas = [(corpus.html_slice(size, i) if kind == "html" else corpus.skewed_bytes(size, i) if kind == "skewed" else corpus.zipf_text(size, i % 64)) for i in range(64)]
if kind == "fixed":
    zs = []
    for d in datas:
        co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        zs.append(co.compress(d) + co.flush())
else:
    zs = [zlib.compress(d, 6) for d in datas]
streams = [zs[i % 64] for i in range(nstreams)]
in_off = np.zeros(nstreams, np.uint64); out_off = np.zeros(nstreams, np.uint64)
ip = op = 0
for k, s in enumerate(streams):
    in_off[k] = ip; out_off[k] = op; ip += (len(s) + 255) // 256 * 256; op += (size + 255) // 256 * 256
in_buf = np.zeros(ip + 16, np.uint8)
for k, s in enumerate(streams):
    in_buf[int(in_off[k]):int(in_off[k]) + len(s)] = np.frombuffer(s, np.uint8)
out_buf = np.zeros(op + 16, np.uint8)
in_len = np.array([len(s) for s in streams], np.uint64); cap = np.full(nstreams, size, np.uint64)
L.pzg_prof_buffer(ctx.handle, nstreams, None)
for rep in range(2):
    out_len, status, detail, in_used, adler = ctx.decompress_many_raw(in_buf, in_off, in_len, out_buf, out_off, cap)
ms = ctx.last_kernel_ms()
prof = np.zeros((nstreams, 16), np.uint64)
L.pzg_prof_buffer(ctx.handle, nstreams, prof.ctypes.data)
m = prof.astype(np.float64).mean(axis=0)
print(f"streams {nstreams} x {size} B; kernel {ms:.3f} ms; status ok {int((status==0).sum())}; outputs ok {all(out_buf[int(out_off[k]):int(out_off[k])+size].tobytes()==datas[k%64] for k in range(min(nstreams,64)))}")
if os.environ.get("PZG_PROF_HDR"):
    hdr = ["  hdr: HCLEN + code-length table", "  hdr: code lengths", "  hdr: literal/length table", "  hdr: distance table"]
names = ["total", "header+tables", "token loop", "flush+adler", "window_append", "checked steps", "#windows", "#tokens queued",
         "  emit: complete_pending", "  emit: scan+stop", "  complete_pending: wait for far bytes", "  emit: far", "emit_segment", "#segments", "#general copies", "#checked steps"]
if os.environ.get("PZG_PROF_HDR"):
    names[8:12] = hdr
if os.environ.get("PZG_PROF_HOT"):  # -DPZG_PROFILE_HOT build
    names[8:12] = ["  hot loop: windows / strips: phase A", "  hot loop: segments / strips: phase B", "  rare window path / strips: compaction", "  strips: emission"]
    names[12:16] = ["    group: records wait + placement + refill issue", "    group: loads issued, classification", "    group: near matches, round 2a", "    group: literal runs (wait + stores)"]
    names[7] = "    group: far matches (wait + stores)"
    names[6] = "    group: later rounds"
for i, nme in enumerate(names):
    if nme == "-":
        continue
    extra = f"({100*m[i]/m[0]:5.1f}%)" if not nme.startswith("#") else ""
    print(f"  {nme:22s} {m[i]:12.0f} {extra}")
print(f"  cycles/window {m[4]/max(m[6],1):.0f}; tokens/window {m[7]/max(m[6],1):.2f}; cycles/segment {m[12]/max(m[13]+m[14],1):.0f}; "
      f"bytes/segment {size/max(m[13],1):.1f}; tokens/segment {m[7]/max(m[13]+m[14],1):.2f}; windows/segment {m[6]/max(m[13],1):.2f}")
