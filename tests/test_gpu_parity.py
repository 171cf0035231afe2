"""GPU parity tests: the HIP path, called through the C ABI of include/pzg.h, against the oracle, the
reference's own golden vectors and the pinned generated vectors.  Integer/byte work: bit-exact, no
tolerance anywhere.  Every parity test runs for the pure 32 KiB LDS ring (15) and for hybrid rings."""
import hashlib
import os
import subprocess
import zlib

import numpy as np
import pytest

import corpus
from conftest import REF_CASES, ROOT, read_case
from test_oracle_golden import load_vectors

pytestmark = pytest.mark.gpu
RINGS = [15, 14, 13, 12, 11]  # every shipped instance (pzg_set_option accepts 11..15)


@pytest.fixture(params=RINGS, ids=lambda r: f"ring{r}")
def ctx(gpu_ctx, request):
    gpu_ctx.set_ring_bits(request.param)
    yield gpu_ctx
    gpu_ctx.set_ring_bits(11)


def run_batch(ctx, streams, caps, align=16, gzip=False):
    """Lay the streams out in arenas (each extent `align`-aligned) and call pzg_decompress_many."""
    n = len(streams)
    in_off = np.zeros(n, dtype=np.uint64)
    out_off = np.zeros(n, dtype=np.uint64)
    ip = op = 0
    for k in range(n):
        in_off[k], out_off[k] = ip, op
        ip += (len(streams[k]) + align - 1) // align * align
        op += (caps[k] + align - 1) // align * align
    in_buf = np.zeros(ip + 16, dtype=np.uint8)
    for k, s in enumerate(streams):
        in_buf[int(in_off[k]):int(in_off[k]) + len(s)] = np.frombuffer(s, dtype=np.uint8)
    out_buf = np.full(op + 16, 0xCD, dtype=np.uint8)
    in_len = np.array([len(s) for s in streams], dtype=np.uint64)
    out_cap = np.array(caps, dtype=np.uint64)
    res = ctx.decompress_many_raw(in_buf, in_off, in_len, out_buf, out_off, out_cap, gzip=gzip)
    outs = [out_buf[int(out_off[k]):int(out_off[k]) + min(int(res[0][k]), caps[k])].tobytes() for k in range(n)]
    return res, outs, out_buf, out_off


@pytest.mark.parametrize("name", REF_CASES)
def test_reference_golden_decompress(ctx, name):
    """test/Test.hs:83-86: assertEqual (Right gold) (decompress z), through the mirror API."""
    import pure_zlib_amd as P
    z, gold = read_case(name)
    assert P.decompress(z, ctx=ctx) == P.Right(gold)


def test_reference_golden_batch(ctx):
    zs, golds = zip(*[read_case(n) for n in REF_CASES])
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, list(zs), [len(g) for g in golds])
    for k, name in enumerate(REF_CASES):
        assert status[k] == 0, name
        assert outs[k] == golds[k], name
        assert int(out_len[k]) == len(golds[k]) and int(in_used[k]) == len(zs[k])
        assert int(adler[k]) == zlib.adler32(golds[k])


def test_pinned_generated_vectors(ctx):
    import pure_zlib_amd.zlib as Z
    vs = load_vectors()
    streams = [bytes.fromhex(v["z"]) for v in vs]
    caps = [max(v["out_len"], 1) for v in vs]
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, caps)
    for k, v in enumerate(vs):
        assert status[k] == v["status"], (v["name"], status[k], detail[k])
        if v["status"] == 0:
            assert hashlib.sha256(outs[k]).hexdigest() == v["out_sha256"], v["name"]
            assert int(adler[k]) == v["adler"] and int(in_used[k]) == v["in_used"] and int(out_len[k]) == v["out_len"]
        else:
            err = Z.error_from_status(streams[k], int(status[k]), detail[k])
            assert err.show() == v["message"], (v["name"], err.show(), v["message"])


def test_valid_streams_vs_oracle(ctx, oracle):
    streams, datas = [], []
    for seed in range(400):
        n = [0, 1, 2, 5, 100, 1000, 5000, 40000, 70000, 200000][seed % 10] if seed % 7 == 0 else (seed * 37) % 20000
        d = corpus.mixed_data(n, seed)
        streams.append(corpus.compress_variant(d, seed))
        datas.append(d)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas])
    for k in range(len(streams)):
        r, o = oracle.decompress(streams[k], len(datas[k]))
        assert r.status == 0 and o == datas[k]
        assert status[k] == 0, (k, status[k], detail[k])
        assert outs[k] == datas[k], k
        assert int(adler[k]) == r.adler and int(in_used[k]) == r.in_used and int(out_len[k]) == r.out_len


def test_corrupt_streams_vs_oracle(ctx, oracle):
    streams, caps = [], []
    for seed in range(2000):
        d = corpus.mixed_data((seed * 131) % 3000 + 1, seed)
        z = corpus.corrupt(corpus.compress_variant(d, seed), seed)
        streams.append(z)
        caps.append([len(d), len(d) + 100, 1 << 17][seed % 3])
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, caps)
    import pure_zlib_amd.zlib as Z
    for k in range(len(streams)):
        r, o = oracle.decompress(streams[k], caps[k])
        assert status[k] == r.status, (k, status[k], r.status, r.message.decode())
        if r.status == 0:
            assert outs[k] == o and int(adler[k]) == r.adler and int(in_used[k]) == r.in_used
        elif r.status == 14:
            assert int(out_len[k]) == r.out_len
        else:
            # same constructor class AND same message as the reference (restated by the oracle)
            err = Z.error_from_status(streams[k], int(status[k]), detail[k])
            assert err.show() == r.message.decode(), (k, err.show(), r.message.decode())


def test_strips_valid_and_corrupt_vs_oracle(ctx, oracle):
    """Round 4: long runs of input are decoded by strips (inflate_core.h strip_span(): 64 lanes, one piece of the input each,
    speculative starts verified and repaired, tokens through the wave's scratch).  Streams of several spans -- one block and
    many, every strategy, two-bit codes -- and six corrupted variants of each, in ONE launch (the waves' scratch is reused from
    stream to stream), against the oracle: status, message, in_used, Adler-32, every byte."""
    streams, caps, datas = [], [], []
    for seed in range(48):
        d, z = corpus.strip_case(seed)
        streams.append(z)
        caps.append(len(d))
        datas.append(d)
        for c in range(6):
            streams.append(corpus.corrupt(z, seed * 16 + c))
            caps.append([len(d) + 64, len(d) // 2][c % 2] if c < 4 else len(d))
            datas.append(None)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, caps)
    import pure_zlib_amd.zlib as Z
    for k in range(len(streams)):
        r, o = oracle.decompress(streams[k], caps[k])
        assert status[k] == r.status, (k, status[k], r.status, r.message.decode())
        if r.status == 0:
            assert outs[k] == o and int(adler[k]) == r.adler and int(in_used[k]) == r.in_used and int(out_len[k]) == r.out_len, k
            assert datas[k] is None or o == datas[k]
        elif r.status == 14:
            assert int(out_len[k]) == r.out_len
        else:
            err = Z.error_from_status(streams[k], int(status[k]), detail[k])
            assert err.show() == r.message.decode(), (k, err.show(), r.message.decode())


def _check_against_oracle(oracle, streams, caps, res, outs, datas=None):
    import pure_zlib_amd.zlib as Z
    out_len, status, detail, in_used, adler = res
    for k in range(len(streams)):
        r, o = oracle.decompress(streams[k], caps[k])
        assert status[k] == r.status, (k, status[k], r.status, r.message.decode())
        if r.status == 0:
            assert outs[k] == o and int(adler[k]) == r.adler and int(in_used[k]) == r.in_used and int(out_len[k]) == r.out_len, k
            assert datas is None or datas[k] is None or o == datas[k], k
        elif r.status == 14:
            assert int(out_len[k]) == r.out_len
        else:
            err = Z.error_from_status(streams[k], int(status[k]), detail[k])
            assert err.show() == r.message.decode(), (k, err.show(), r.message.decode())


def test_exotic_streams_vs_oracle(ctx, oracle):
    """Round 5: the strips and the groups had only ever met what zlib's encoder writes.  96 streams from
    tests/deflate_writer.py -- a DEFLATE writer that emits chosen tokens with chosen code lengths: 13-15-bit literal/length and
    distance codes in blocks long enough for spans, incomplete codes, a single distance code, code 16 first, code-length runs past
    HLIT + HDIST, HLIT 288 / HDIST 32, matches of 258 bytes and of distance 32768, distance-1 runs, hundreds of tiny dynamic blocks
    at odd bit offsets, fixed / stored / dynamic interleaved (the system zlib rejects nearly all of them; the reference accepts them:
    Deflate.hs:124-156, HuffmanTree.hs:43-83) -- and five corrupted variants of each: 576 streams in ONE launch per ring, against
    the oracle: status, message, in_used, Adler-32, every byte."""
    import deflate_writer as W
    streams, caps, datas = [], [], []
    for seed in range(96):
        d, z, _ = W.exotic_stream(seed)
        streams.append(z)
        caps.append(len(d))
        datas.append(d)
        for c in range(5):
            streams.append(corpus.corrupt(z, seed * 16 + c))
            caps.append([len(d) + 64, len(d) // 2, len(d)][c % 3])
            datas.append(None)
    res, outs, _, _ = run_batch(ctx, streams, caps)
    _check_against_oracle(oracle, streams, caps, res, outs, datas)


def test_strips_laid_out_by_a_profile_of_another_kind_of_stream(ctx, oracle):
    """Round 5: a stream-wave remembers how the tokens of the last stream's first span were spread (strip_profile_learn) and cuts
    the next stream's strips at those positions -- checked by the run-ups, laid out again in equal strips when they disagree
    (strip_profile_check), forgotten when a lane runs out of steps.  Nothing about the result may depend on any of it: 1,280
    streams whose neighbours differ in kind (text, html, literal-heavy, binary, writer-made), size (8-64 KiB), level and validity,
    launched THREE times (every wave's profile after the first launch is whatever its last stream left there), every launch
    against the oracle: status, message, in_used, Adler-32, every byte."""
    import deflate_writer as W
    rng = np.random.default_rng(11)
    kinds = []
    for i in range(40):
        size = int(rng.choice([8192, 16384, 24576, 32768, 49152, 65536]))
        d = [corpus.zipf_text, corpus.html_slice, corpus.skewed_bytes, corpus.mixed_data][i % 4](size, i)
        kinds.append((d, zlib.compress(d, [6, 1, 9, 6, 3][(i // 4) % 5])))
    for seed in (0, 3, 5, 7):
        d, z, _ = W.exotic_stream(seed)
        kinds.append((d, z))
    streams, caps, datas = [], [], []
    for j in rng.integers(0, len(kinds), 1280):
        d, z = kinds[j]
        bad = rng.integers(0, 8) == 0
        streams.append(corpus.corrupt(z, int(rng.integers(0, 1 << 20))) if bad else z)
        caps.append(len(d))
        datas.append(None if bad else d)
    expect = [oracle.decompress(z, c) for z, c in zip(streams, caps)]
    for launch in range(3):
        res, outs, _, _ = run_batch(ctx, streams, caps)
        out_len, status, detail, in_used, adler = res
        for k, (r, o) in enumerate(expect):
            assert status[k] == r.status, (launch, k, status[k], r.status)
            if r.status == 0:
                assert outs[k] == o and int(adler[k]) == r.adler and int(in_used[k]) == r.in_used, (launch, k)
                assert datas[k] is None or o == datas[k], (launch, k)
    _check_against_oracle(oracle, streams[:200], caps[:200], (out_len[:200], status[:200], detail[:200], in_used[:200], adler[:200]), outs[:200], datas[:200])


def test_strips_with_failing_guesses_on_the_device(tmp_path):
    """the "guesses keep failing" path of the strips (spans that end after a strip or two, `poor`, the rest
    of the block left to the windows; repaired lanes re-storing their regions) had only run on the one-lane host model, which
    cannot see a cross-lane ordering bug.  build/lab_poor/libpzg.so is the product's source with the run-up cut to 8 bits and two
    rounds of phase B (tests/tools/lab_build.sh; built by tests/test_exotic_streams.py on the CPU side): zlib-made and writer-made
    streams, valid and corrupted, rings 11 and 15, against the oracle -- in a child process (PZG_LIB)."""
    import sys
    from test_exotic_streams import POOR_FLAGS, lab_library
    so = lab_library("poor", POOR_FLAGS)
    code = r'''
import os, sys, zlib
sys.path.insert(0, os.path.join(os.environ["PZG_ROOT"], "tests")); sys.path.insert(0, os.environ["PZG_ROOT"])
import torch; torch.cuda.init()
import corpus, deflate_writer as W
import pure_zlib_amd as P
from pure_zlib_amd import _ffi
from oracle import oracle as O
assert _ffi.LIB_PATH.endswith("build/lab_poor/libpzg.so"), _ffi.LIB_PATH
ctx = P.Context(0)
streams = []
for seed in range(40):
    z = W.exotic_stream(seed)[1] if seed % 2 else corpus.strip_case(seed)[1]
    streams.append(z)
    for c in range(3):
        streams.append(corpus.corrupt(z, seed * 16 + c))
for ring in (11, 15):
    ctx.set_ring_bits(ring)
    got = P.decompress_many(streams, ctx=ctx)
    for k, (z, g) in enumerate(zip(streams, got)):
        r, o = O.decompress(z, 1 << 21)
        if r.status == 0:
            assert g == P.Right(o), (ring, k)
        else:
            assert (not g.is_right()) and g.value.show() == r.message.decode(), (ring, k, g, r.message)
ctx.close()
print("poor strips parity ok", len(streams))
'''
    env = dict(os.environ, PZG_LIB=so, PZG_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "poor strips parity ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])


def test_strips_laid_out_by_fuzzed_profiles_on_the_device(tmp_path):
    """the wave's profile lives in library scratch that outlives streams and launches, and steers how a span's strips
    are cut.  build/lab_profseed/libpzg.so (tests/tools/lab_build.sh -DPZG_LAB -DPZG_LAB_SEED_PROFILE; built on the CPU side by
    tests/test_exotic_streams.py) overwrites the profile words before EVERY stream with a well-marked profile that no stream taught the
    wave -- random, unsorted, decreasing, constant, squeezed, stretched, with odd extents and token counts: text, html and literal-heavy
    streams of 20-70 KiB (the ones a profile is consulted for), writer-made and strip-case streams, valid and corrupted, rings 11 and 15,
    three launches, against the oracle -- in a child process (PZG_LIB)."""
    import sys
    from test_exotic_streams import PROFSEED_FLAGS, lab_library
    so = lab_library("profseed", PROFSEED_FLAGS)
    code = r'''
import os, sys, zlib
sys.path.insert(0, os.path.join(os.environ["PZG_ROOT"], "tests")); sys.path.insert(0, os.environ["PZG_ROOT"])
import torch; torch.cuda.init()
import corpus, deflate_writer as W
import pure_zlib_amd as P
from pure_zlib_amd import _ffi
from oracle import oracle as O
assert _ffi.LIB_PATH.endswith("build/lab_profseed/libpzg.so"), _ffi.LIB_PATH
ctx = P.Context(0)
streams = []
for seed in range(160):
    n = [20000, 32768, 50000, 70000][seed % 4]
    d = [corpus.zipf_text, corpus.html_slice, corpus.skewed_bytes][seed % 3](n, seed)
    streams.append(zlib.compress(d, [6, 1, 9][seed % 3]))
for seed in range(24):
    z = W.exotic_stream(seed)[1] if seed % 2 else corpus.strip_case(seed)[1]
    streams.append(z)
    streams.append(corpus.corrupt(z, seed * 16))
exp = [O.decompress(z, 1 << 21) for z in streams]
for ring in (11, 15):
    ctx.set_ring_bits(ring)
    for launch in range(3):
        got = P.decompress_many(streams, ctx=ctx)
        for k, (g, (r, o)) in enumerate(zip(got, exp)):
            if r.status == 0:
                assert g == P.Right(o), (ring, launch, k)
            else:
                assert (not g.is_right()) and g.value.show() == r.message.decode(), (ring, launch, k, g, r.message)
ctx.close()
print("fuzzed profiles parity ok", len(streams))
'''
    env = dict(os.environ, PZG_LIB=so, PZG_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "fuzzed profiles parity ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])


def test_text_blobs_with_far_back_references(ctx, oracle):
    """32-100 KiB Zipf text: distances up to 32 KiB, i.e. older than every hybrid ring."""
    streams, datas = [], []
    for seed in range(48):
        n = [32768, 65536, 100000, 4096][seed % 4]
        d = corpus.zipf_text(n, seed)
        datas.append(d)
        streams.append(zlib.compress(d, 1 + seed % 9))
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas])
    for k in range(len(streams)):
        assert status[k] == 0 and outs[k] == datas[k] and int(adler[k]) == zlib.adler32(datas[k]), k


def test_far_reads_next_to_a_neighbours_extent(ctx):
    """output extents back to back at odd addresses (no 128-byte alignment anywhere), every stream with matches
    older than the small rings: a far read's cache line may also hold the first or last bytes of the NEIGHBOURING stream's
    extent, which another wave is still writing.  Only a stream's own bytes may ever be used."""
    streams, datas = [], []
    for seed in range(96):
        d = corpus.zipf_text(5000 + 977 * (seed % 37), 300 + seed)
        datas.append(d)
        streams.append(zlib.compress(d, 1 + seed % 9))
    for rep in range(3):  # (which waves run side by side changes from launch to launch)
        (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas], align=1)
        for k in range(len(streams)):
            assert status[k] == 0 and outs[k] == datas[k] and int(adler[k]) == zlib.adler32(datas[k]), (rep, k)


def test_output_never_written_past_capacity(ctx):
    d = corpus.zipf_text(50000, 3)
    z = zlib.compress(d, 6)
    caps = [0, 1, 15, 16, 17, 4095, 4096, 32768, 49999]
    (out_len, status, detail, in_used, adler), outs, out_buf, out_off = run_batch(ctx, [z] * len(caps), caps)
    for k, cap in enumerate(caps):
        assert status[k] == 14 and int(out_len[k]) == len(d)
        assert outs[k] == d[:cap]
        lo = int(out_off[k]) + cap
        hi = int(out_off[k + 1]) if k + 1 < len(caps) else lo
        assert (out_buf[lo:hi] == 0xCD).all()  # padding between extents untouched


def test_small_capacity_and_corrupt_trailer(ctx, oracle):
    """An output larger than its capacity AND a bad checksum: the reference outcome is the checksum error."""
    d = corpus.zipf_text(40000, 8)
    z = bytearray(zlib.compress(d, 6))
    z[-1] ^= 1
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, [bytes(z)] * 3, [100, 20000, 40000])
    r, _ = oracle.decompress(bytes(z), 100)
    assert r.status == 10 and list(status) == [10, 10, 10]
    assert tuple(detail[0]) == (r.detail0, r.detail1)


def test_unaligned_extents(ctx):
    streams, datas = [], []
    for seed in range(64):
        d = corpus.mixed_data(1000 + seed * 97, seed + 11)
        datas.append(d)
        streams.append(zlib.compress(d, 1 + seed % 9))
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas], align=1)
    for k in range(len(streams)):
        assert status[k] == 0 and outs[k] == datas[k], k


def test_mirror_api_chunks_and_retry(gpu_ctx):
    """Zlib.hs:37-51 through the mirror: chunked lazy ByteStrings, 'Finished with data remaining.',
    trailing bytes ignored, outputs larger than the first capacity guess (relaunch with the exact size)."""
    import pure_zlib_amd as P
    d = corpus.zipf_text(300000, 2)
    z = zlib.compress(d, 9)
    assert P.decompress(z, ctx=gpu_ctx) == P.Right(d)  # 300 KB out of ~90 KB in: needs the retry
    assert P.decompress([z[:1000], z[1000:50000], z[50000:]], ctx=gpu_ctx) == P.Right(d)
    assert P.decompress(z + b"tail", ctx=gpu_ctx) == P.Right(d)
    got = P.decompress([z, b"tail"], ctx=gpu_ctx)
    assert got == P.Left(P.DecompressionError_("Finished with data remaining."))
    assert P.decompress(b"", ctx=gpu_ctx) == P.Left(P.DecompressionError_("Ran out of data mid-decompression 2."))
    many = P.decompressMany([z, z[:-5], b"\x78\x9d", zlib.compress(b"ok")], ctx=gpu_ctx)
    assert many[0] == P.Right(d) and many[3] == P.Right(b"ok")
    assert many[1].value.show() == "Decompression error: Ran out of data mid-decompression 2."
    assert many[2].value.show() == "Header error: Header checksum failed"


def test_batch_8192_level6_blobs_every_stream_checked(ctx):
    """BASELINE config 4 shape (32 KiB level-6 blobs) at 1/8 of the stream count: every stream's status,
    length, in_used and Adler-32; bench.py repeats this at the full 65,536 with a full byte compare."""
    pool = [corpus.zipf_text(32768, s) for s in range(64)]
    zs = [zlib.compress(t, 6) for t in pool]
    ad = [zlib.adler32(t) for t in pool]
    n = 8192
    pick = np.random.default_rng(3).integers(0, 64, size=n)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, [zs[i] for i in pick], [32768] * n, align=256)
    assert (status == 0).all() and (out_len == 32768).all()
    assert (adler == np.array([ad[i] for i in pick], dtype=np.uint32)).all()
    assert (in_used == np.array([len(zs[i]) for i in pick], dtype=np.uint64)).all()
    for k in range(0, n, 257):
        assert outs[k] == pool[pick[k]]


def test_adler32_kernel(gpu_ctx):
    rng = np.random.default_rng(5)
    for n in [0, 1, 15, 16, 17, 1000, 65535, 65536, 65537, 1 << 20, (1 << 22) + 12345, 50_000_000, 600_000_011]:
        buf = rng.integers(0, 256, size=n + 32, dtype=np.uint8)
        for skew in (0, 3):
            view = buf[skew:skew + n]
            assert gpu_ctx.adler32(view) == zlib.adler32(view.tobytes()), (n, skew)
    worst = np.full(3_000_000, 255, dtype=np.uint8)
    assert gpu_ctx.adler32(worst) == zlib.adler32(worst.tobytes())
    assert gpu_ctx.adler32(worst[:70000], init=0xFFF0FFF0) == zlib.adler32(worst[:70000].tobytes(), 0xFFF0FFF0)


def test_cxx_abi_smoke_binary():
    """The C ABI from plain C++ (no Python in the loop): tests/cxx/abi_smoke.cpp over the nine fixtures."""
    exe = os.path.join(ROOT, "tests", "cxx", "abi_smoke")
    src = os.path.join(ROOT, "tests", "cxx", "abi_smoke.cpp")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-o", exe, "-L" + os.path.join(ROOT, "pure_zlib_amd"),
                               "-lpzg", "-Wl,-rpath," + os.path.join(ROOT, "pure_zlib_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    args = []
    for n in REF_CASES:
        args += [os.path.join(ROOT, "tests", "golden", "ref", n + ".z"), os.path.join(ROOT, "tests", "golden", "ref", n + ".gold")]
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" OK ") == len(REF_CASES)


def test_incremental_protocol_and_cli(gpu_ctx, tmp_path, capsys):
    """ZlibDecoder protocol (Monad.hs:163-167) and the deflate CLI (Deflate.hs:15-48)."""
    from pure_zlib_amd import deflate_cli
    from pure_zlib_amd.incremental import Chunk, DecompError, Done, NeedMore, decompress_incremental
    d = corpus.zipf_text(150000, 4)
    z = zlib.compress(d, 6)
    st = decompress_incremental(gpu_ctx)
    assert isinstance(st, NeedMore)
    st = st.feed(b"")
    assert isinstance(st, NeedMore)
    got, pieces = [], [z[i:i + 7000] for i in range(0, len(z), 7000)]
    for p in pieces:
        st = st.feed(p)
        while isinstance(st, Chunk):  # chunks come out as soon as 64 KiB are buffered, not at the end
            got.append(st.chunk)
            st = st.next()
        assert isinstance(st, NeedMore) or p is pieces[-1]
    assert isinstance(st, Done) and b"".join(got) == d
    assert [len(c) for c in got[:-1]] == [32768] * (len(got) - 1) and 32768 <= len(got[-1]) < 65536
    bad = decompress_incremental(gpu_ctx).feed(b"\x78\x9d\x00")
    assert isinstance(bad, DecompError) and bad.error.show() == "Header error: Header checksum failed"
    # the CLI
    src = tmp_path / "blob.z"
    src.write_bytes(z)
    assert deflate_cli.main([str(src)]) == 0
    assert (tmp_path / "blob").read_bytes() == d
    deflate_cli.main([str(tmp_path / "blob.txt")])
    deflate_cli.main([])
    (tmp_path / "cut.z").write_bytes(z[:-9])
    deflate_cli.main([str(tmp_path / "cut.z")])
    out = capsys.readouterr().out
    assert "Unexpected file name." in out and "USAGE: deflate [filename]" in out
    assert "ERROR: Ran out of data mid-decompression." in out
    # batch mode: several files, ONE decompressMany call; per file what `decompress` returns
    datas = [corpus.zipf_text(3000 + 7919 * k, 40 + k) for k in range(5)]
    names = []
    for k, dk in enumerate(datas):
        (tmp_path / f"m{k}.z").write_bytes(zlib.compress(dk, 1 + k))
        names.append(str(tmp_path / f"m{k}.z"))
    (tmp_path / "bad.z").write_bytes(zlib.compress(datas[0], 6)[:-5])
    (tmp_path / "tail.z").write_bytes(zlib.compress(datas[1], 6) + b"x" * 40000)  # a whole unread lazy chunk behind the stream
    assert deflate_cli.main(["--many"] + names + [str(tmp_path / "bad.z"), str(tmp_path / "note.txt"), str(tmp_path / "tail.z")]) == 0
    for k, dk in enumerate(datas):
        assert (tmp_path / f"m{k}").read_bytes() == dk
    deflate_cli.main(names[:2])  # two arguments without the flag: the reference's usage line (Deflate.hs:17-29)
    out = capsys.readouterr().out
    assert out.count("USAGE: deflate [filename]") == 1
    assert "bad.z: ERROR: Decompression error: Ran out of data mid-decompression 2." in out
    assert "note.txt: Unexpected file name." in out
    assert "tail.z: ERROR: Decompression error: Finished with data remaining." in out
    assert not (tmp_path / "bad").exists() and not (tmp_path / "tail").exists()


def test_benchmark_harness_groups(gpu_ctx, capsys):
    """the criterion groups of Benchmark.hs:26-46 plus the batch sweep, every sample checked against .gold."""
    from pure_zlib_amd import benchmark
    bs = benchmark.build_benchmarks(benchmark.DEFAULT_DIR, ["rfctest1", "zerotest2"], [1, 64], ctx=gpu_ctx)
    names = [b[0] for b in bs]
    assert names[:4] == ["decompression/rfctest1/normal/pzgpu", "decompression/rfctest1/normal/zlib",
                         "decompression/rfctest1/incremental/pzgpu", "decompression/rfctest1/incremental/zlib"]
    assert "batch/zerotest2/n=64/pzgpu" in names
    for _name, thunk, nbytes in bs:
        thunk()  # raises if any output differs from the gold file
        assert nbytes > 0
    assert benchmark.main(["--cases", "randtest1", "--time-limit", "0.05", "--batch", "8"]) == 0
    out = capsys.readouterr().out
    assert out.count("benchmarking ") == 5 and "full output checked" in out


def test_long_codes_second_level_tables(ctx, oracle):
    """Literal-heavy and html corpora: literal/length codes longer than the 8-bit primary table (second-level
    tables in the windows, the exact walk in the checked path), valid and corrupted, against zlib and the oracle."""
    streams, datas = [], []
    for seed in range(600):
        n = [700, 3000, 20000, 33000, 70000][seed % 5]
        d = corpus.skewed_bytes(n, seed) if seed % 2 else corpus.html_slice(n, seed)
        streams.append(corpus.compress_variant(d, seed) if seed % 3 == 0 else zlib.compress(d, 1 + seed % 9))
        datas.append(d)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas])
    for k in range(len(streams)):
        assert status[k] == 0 and outs[k] == datas[k], (k, status[k], detail[k])
        assert int(adler[k]) == zlib.adler32(datas[k]) and int(in_used[k]) == len(streams[k])
    bad = [corpus.corrupt(streams[k % 600], 9000 + k) for k in range(1500)]
    caps = [len(datas[k % 600]) + 64 for k in range(1500)]
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, bad, caps)
    import pure_zlib_amd.zlib as Z
    for k in range(len(bad)):
        r, o = oracle.decompress(bad[k], caps[k])
        assert int(status[k]) == r.status, (k, status[k], r.status, r.message)
        if r.status == 0:
            assert outs[k] == o and int(adler[k]) == r.adler and int(in_used[k]) == r.in_used
        elif r.status == 14:
            assert int(out_len[k]) == r.out_len
        else:
            err = Z.error_from_status(bad[k], int(status[k]), detail[k])
            assert err.show() == r.message.decode(), (k, err.show(), r.message.decode())


def test_binary_records_long_codes_inside_spans(ctx, oracle):
    """Round 6: 16-byte binary-looking records (corpus.binary_records: ~250 literals of 8 to 10 bits, more second-level entries than
    the pool holds, a token every ~200 that only decode_long() resolves).  The spans' lanes take those tokens through strip_resolve()
    instead of stopping; what the depth cap leaves of the pool goes to the first prefixes it cut short.  All three levels, 8-64 KiB,
    valid and corrupted, launched twice (the second launch meets the waves' profiles of the first): every byte against the oracle."""
    streams, datas = [], []
    for seed in range(400):
        d = corpus.binary_records([8, 16, 30, 64, 33][seed % 5] * 1024, seed)
        streams.append(zlib.compress(d, [6, 1, 9][seed % 3]))
        datas.append(d)
    for launch in range(2):
        (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas])
        for k in range(len(streams)):
            assert status[k] == 0 and outs[k] == datas[k], (launch, k, status[k], detail[k])
            assert int(adler[k]) == zlib.adler32(datas[k]) and int(in_used[k]) == len(streams[k])
    bad = [corpus.corrupt(streams[k % 400], 7000 + k) for k in range(1200)]
    caps = [len(datas[k % 400]) + (64 if k % 3 else 0) for k in range(1200)]
    res, outs, _, _ = run_batch(ctx, bad, caps)
    _check_against_oracle(oracle, bad, caps, res, outs, [None] * len(bad))


def test_codes_around_the_second_level_pool(ctx, oracle):
    """... and writer-made codes of 150-256 literals of 8 to 13 bits (deflate_writer.pool_stream: a little less, a little more and much
    more than the pool's 252 second-level entries; one to three blocks, every kind of distance), 240 valid streams and 720 corrupted ones
    in one launch, against the oracle."""
    import deflate_writer as W
    streams, caps, datas = [], [], []
    for seed in range(240):
        d, z = W.pool_stream(seed)
        streams.append(z); caps.append(len(d)); datas.append(d)
        for k in range(3):
            streams.append(corpus.corrupt(z, 8 * seed + k)); caps.append([len(d), len(d) + 100, len(d) // 2][k]); datas.append(None)
    res, outs, _, _ = run_batch(ctx, streams, caps)
    _check_against_oracle(oracle, streams, caps, res, outs, datas)


def test_cxx_module_mirror_reads_like_the_reference_tests():
    """The C++ host mirror of Codec.Compression.Zlib (pure_zlib_amd/cxx/codec_compression_zlib.hpp) driven by
    tests/cxx/test_mirror.cpp: Test.hs's nine cases, decompressMany, the chunk rule, error values, the incremental decoder."""
    exe = os.path.join(ROOT, "tests", "cxx", "test_mirror")
    src = os.path.join(ROOT, "tests", "cxx", "test_mirror.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", src, "-o", exe, "-L" + os.path.join(ROOT, "pure_zlib_amd"),
                           "-lpzg", "-Wl,-rpath," + os.path.join(ROOT, "pure_zlib_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "ref")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(" OK") == 18 and "0 failure(s)" in out.stdout


def test_gzip_members_extension(ctx, oracle):
    """SURVEY 8f row 4 (an extension: the reference has no gzip): RFC 1952 members through PZG_GZIP -- same DEFLATE
    kernel, gzip header forms, CRC-32 (per-lane accumulators over coalesced 1 KiB blocks) and ISIZE verified on the device -- against
    system zlib (wbits 31) for valid members of many sizes and the oracle for corrupted ones."""
    import pure_zlib_amd as P
    streams, datas = [], []
    # (the CRC kernel works in 1 KiB blocks of 64 x 16 bytes, front-padded: lengths around every boundary of both)
    sizes = [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 1007, 1008, 1009, 1023, 1024, 1025, 2047, 2048,
             2049, 4093, 4096, 33000, 70001, 262144 + 3, 1 << 20]
    for seed in range(300):
        n = sizes[(seed // 3) % len(sizes)] if seed % 3 == 0 else (seed * 131) % 20000
        d = corpus.mixed_data(n, seed) if seed % 2 else corpus.html_slice(min(n, 100000), seed)
        streams.append(corpus.gzip_member(d, seed))
        datas.append(d)
    (out_len, status, detail, in_used, crc), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas], gzip=True)
    for k in range(len(streams)):
        assert zlib.decompress(streams[k], 31) == datas[k]
        assert status[k] == 0 and outs[k] == datas[k], (k, len(datas[k]), status[k], detail[k])
        assert int(crc[k]) == zlib.crc32(datas[k]) and int(in_used[k]) == len(streams[k]) and int(out_len[k]) == len(datas[k])
    bad = [corpus.corrupt(streams[k % 300], 5000 + k) for k in range(1200)]
    caps = [len(datas[k % 300]) + 4096 for k in range(1200)]
    (out_len, status, detail, in_used, crc), outs, _, _ = run_batch(ctx, bad, caps, gzip=True)
    for k in range(len(bad)):
        r, o = oracle.gzip_decompress(bad[k], caps[k])
        if int(status[k]) == 14 and r.status in (10, 19):
            assert int(out_len[k]) == r.out_len  # documented: CRC/ISIZE of an unstored output are not verified
            continue
        assert int(status[k]) == r.status, (k, status[k], r.status, r.message)
        if r.status == 0:
            assert outs[k] == o and int(crc[k]) == r.adler
        elif r.status in (10, 19):
            assert [int(detail[k][0]), int(detail[k][1])] == [r.detail0, r.detail1]
    # the mirror API
    rs = P.gzip_decompress_many(streams[:20], ctx=ctx)
    assert rs == [P.Right(d) for d in datas[:20]]
    e = P.gzip_decompress_many([b"\x1f\x8c" + streams[1][2:]], ctx=ctx)[0]
    assert isinstance(e, P.Left) and e.value.show() == "Header error: gzip: bad magic"


def test_large_stream_and_many_tiny_streams(ctx):
    """Scale edges: one 48 MiB multi-block stream on a single wavefront (64-bit cursors, thousands of flushes,
    far reads across the whole run) and 200,000 tiny streams in one launch (the persistent waves' queue)."""
    big = corpus.html_slice(100000, 3) * 300 + corpus.skewed_bytes(1 << 20, 5) + corpus.zipf_text(16 << 20, 9)
    co = zlib.compressobj(6)
    zbig = b"".join(co.compress(big[i:i + (1 << 20)]) + co.flush(zlib.Z_FULL_FLUSH if i % 3 == 0 else zlib.Z_SYNC_FLUSH)
                    for i in range(0, len(big), 1 << 20)) + co.flush()
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, [zbig], [len(big)])
    assert status[0] == 0 and int(out_len[0]) == len(big) and int(adler[0]) == zlib.adler32(big) and outs[0] == big
    tiny_d = [bytes([k % 251]) * (k % 40) + str(k).encode() for k in range(200000)]
    tiny_z = [zlib.compress(d, 1 + k % 9) for k, d in enumerate(tiny_d)]
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, tiny_z, [len(d) for d in tiny_d])
    assert (status == 0).all() and all(o == d for o, d in zip(outs, tiny_d))
