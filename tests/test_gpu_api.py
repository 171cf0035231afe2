"""GPU tests of the C ABI beyond the single decode call: in-library multi-device sharding, the
device-side launch order, the batched Adler-32 (BASELINE config 2's second form), argument validation, and
several host threads on one context."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

import corpus
from conftest import REF_CASES, ROOT
from test_gpu_parity import run_batch

pytestmark = pytest.mark.gpu


def _shard_devices(k):
    """The devices of a k-shard test context: the box's real devices when it has several (shard j on device j mod count --
    a hipSetDevice / peer / pinned-memory mistake then shows), all k shards on device 0 on a one-GPU box."""
    import torch
    nd = torch.cuda.device_count()
    return [j % nd for j in range(k)]


def _mixed_batch(n):
    datas = [corpus.zipf_text(200 + (k * 7919) % 60000, k) if k % 3 else corpus.mixed_data((k * 131) % 9000, k) for k in range(n)]
    return [zlib.compress(d, 1 + k % 9) for k, d in enumerate(datas)], datas


def test_decompress_many_sharded_over_a_device_mask(gpu_ctx, oracle, monkeypatch):
    """pzg_init_mask / pzg_init_devices: ONE pzg_decompress_many call partitions the streams over the context's shards
    (four here: the box's own devices when it has several, all four on device 0 otherwise, so the multi-shard path -- one
    host thread, stream set and arenas per shard, results written straight into the caller's slots -- runs on a 1-GPU
    box and upgrades itself on a node).  Byte-identical to the single-device result and to the oracle."""
    import pure_zlib_amd as P
    import torch
    group = P.Context(devices=_shard_devices(4))
    if torch.cuda.device_count() >= 2:  # the mask form over every visible device too
        allm = P.Context(device_mask=0)
        assert allm.device_count == torch.cuda.device_count()
        streams0, datas0 = _mixed_batch(200)
        assert P.decompress_many(streams0, ctx=allm) == [P.Right(d) for d in datas0]
        allm.close()
    try:
        assert group.device_count == 4 and gpu_ctx.device_count == 1
        streams, datas = _mixed_batch(3000)
        streams += [streams[5][:-7], b"\x78\x9d\x01", streams[9][:40] + b"\xff" * 9 + streams[9][49:]]  # and a few bad ones
        datas += [b"", b"", b""]
        caps = [len(d) + 16 for d in datas]
        one, outs1, _, _ = run_batch(gpu_ctx, streams, caps)
        many, outsN, _, _ = run_batch(group, streams, caps)
        for a, b in zip(one, many):
            assert np.array_equal(a, b)
        assert outs1 == outsN
        for k in range(0, len(streams), 37):
            r, o = oracle.decompress(streams[k], caps[k])
            assert int(many[1][k]) == r.status
            if r.status == 0:
                assert outsN[k] == o and int(many[4][k]) == r.adler
        # the mirror API on the group context
        rs = P.decompress_many(streams[:200], ctx=group)
        assert rs == [P.Right(d) for d in datas[:200]]
        # device pointers belong to one device: refused on a multi-device context
        from pure_zlib_amd._ffi import PzgError
        with pytest.raises(PzgError):
            group.decompress_many_device(1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1)
    finally:
        group.close()


def test_device_side_launch_order(gpu_ctx):
    """PZG_LPT_ORDER: the longest-first launch permutation built on the device changes nothing but the order."""
    from devbatch import DeviceBatch
    texts = [corpus.zipf_text(1024 * (1 + (s * 2654435761 >> 7) % 64), s) for s in range(256)]
    zs = [zlib.compress(t, 6) for t in texts]
    pick = np.random.default_rng(11).integers(0, len(zs), size=20000)
    b = DeviceBatch(texts, zs, pick)
    res = b.run(gpu_ctx, 11)
    b.check_all(*res)
    import torch
    b.d_out.fill_(0xCD)
    b.d_status.fill_(-1)
    torch.cuda.synchronize()
    gpu_ctx.decompress_many_device(b.d_in.data_ptr(), b.d_in_off.data_ptr(), b.d_in_len.data_ptr(), b.d_out.data_ptr(),
                                   b.d_out_off.data_ptr(), b.d_out_cap.data_ptr(), b.d_out_len.data_ptr(), b.d_status.data_ptr(),
                                   b.d_detail.data_ptr(), b.d_in_used.data_ptr(), b.d_adler.data_ptr(), b.n, sync=True, lpt=True)
    b.check_all(b.d_status.cpu().numpy(), b.d_out_len.cpu().numpy(), b.d_in_used.cpu().numpy(), b.d_adler.cpu().numpy().view(np.uint32))


def test_adler32_many_batched(gpu_ctx):
    """BASELINE config 2, batched form: one wave per buffer; every length class, every 16-byte alignment."""
    import torch
    rng = np.random.default_rng(2)
    lens = [0, 1, 15, 16, 17, 255, 4096, 65535, 65536, 65537, 100001, 1 << 20] * 20 + [65536] * 2048
    off, pos = [], 0
    for k, n in enumerate(lens):
        pos += k % 16 if k < 400 else 0  # unaligned starts
        off.append(pos)
        pos += n
    buf = rng.integers(0, 256, size=pos + 64, dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d_buf = torch.from_numpy(buf).to(dev)
    d_off = torch.tensor(off, dtype=torch.int64, device=dev)
    d_len = torch.tensor(lens, dtype=torch.int64, device=dev)
    d_out = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    gpu_ctx.adler32_many_device(d_buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), d_out.data_ptr(), len(lens))
    got = d_out.cpu().numpy().view(np.uint32)
    for k, n in enumerate(lens):
        assert int(got[k]) == zlib.adler32(buf[off[k]:off[k] + n].tobytes()), (k, n)


def test_argument_validation(gpu_ctx):
    """Wrapped extents and NULL outputs are refused or handled, never followed."""
    L = gpu_ctx._L
    z = zlib.compress(b"hello world" * 100)
    zb = np.frombuffer(z, dtype=np.uint8)
    out = np.zeros(4096, dtype=np.uint8)
    u64 = lambda *v: np.array(v, dtype=np.uint64)  # noqa: E731
    olen, st = u64(0), np.array([0], dtype=np.int32)

    def call(in_off, in_len, out_off, out_cap, outp=out):
        return L.pzg_decompress_many(gpu_ctx.handle, zb.ctypes.data, in_off.ctypes.data, in_len.ctypes.data,
                                     outp.ctypes.data if outp is not None else None, out_off.ctypes.data, out_cap.ctypes.data,
                                     olen.ctypes.data, st.ctypes.data, None, None, None, 1, 0)
    assert call(u64(0), u64(len(z)), u64(0), u64(4096)) == 0 and st[0] == 0 and olen[0] == 1100
    assert call(u64(2**64 - 8), u64(64), u64(0), u64(4096)) == -1          # in_off + in_len wraps
    assert call(u64(0), u64(len(z)), u64(2**64 - 16), u64(4096)) == -1      # out_off + out_cap wraps
    assert call(u64(0), u64(len(z)), u64(0), u64(4096), None) == -1         # capacity but no output pointer
    # the single-stream form with no output buffer: counts, stores nothing
    ol, s2, used = C.c_uint64(0), C.c_int32(-1), C.c_uint64(0)
    rc = L.pzg_decompress(gpu_ctx.handle, zb.ctypes.data, len(z), None, 12345, C.byref(ol), C.byref(s2), None, C.byref(used))
    assert rc == 0 and s2.value == 14 and ol.value == 1100


def test_many_tiny_streams_through_the_mirror_without_size_hints(gpu_ctx):
    """200,000 tiny streams and no size hints: the first-pass capacities stay small (the batch is packed, not the
    covering range of 64 KiB slots), streams that need more are relaunched with their exact size."""
    import pure_zlib_amd as P
    datas = [str(k).encode() * (1 + k % 9) for k in range(200000)]
    datas[777] = corpus.zipf_text(300000, 1)  # one that outgrows any first guess
    zs = [zlib.compress(d, 6) for d in datas]
    rs = P.decompress_many(zs, ctx=gpu_ctx)
    assert all(r == P.Right(d) for r, d in zip(rs, datas))


def test_cxx_threads_overlap_on_one_context():
    """SURVEY 8b "Threading": eight host threads on ONE context -- every result right, and the calls overlap (each takes
    its own pipeline of the context) instead of queueing behind one lock."""
    exe = os.path.join(ROOT, "tests", "cxx", "threads")
    src = os.path.join(ROOT, "tests", "cxx", "threads.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", src, "-o", exe, "-L" + os.path.join(ROOT, "pure_zlib_amd"),
                           "-lpzg", "-Wl,-rpath," + os.path.join(ROOT, "pure_zlib_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    args = []
    for n in REF_CASES:
        args += [os.path.join(ROOT, "tests", "golden", "ref", n + ".z"), os.path.join(ROOT, "tests", "golden", "ref", n + ".gold")]
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "160 calls from 8 threads, 0 bad" in out.stdout, out.stdout + out.stderr
    speedup = float(out.stdout.split("speed-up over one thread:")[1].split("x")[0])
    assert speedup > 1.15, out.stdout


def _build_cxx(name, extra=()):
    exe = os.path.join(ROOT, "tests", "cxx", name)
    src = os.path.join(ROOT, "tests", "cxx", name + ".cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", *extra, src, "-o", exe, "-L" + os.path.join(ROOT, "pure_zlib_amd"),
                           "-lpzg", "-Wl,-rpath," + os.path.join(ROOT, "pure_zlib_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_decoder_and_context_lifetimes_in_any_order():
    """include/pzg.h "Lifetimes" (the reference's ZlibDecoder is a GC'd closure, Monad.hs:163-197: it can be dropped at any
    time, in any order): decoder before context, context before decoder, a decoder USED after pzg_shutdown, the
    invalidated handle refused.  Round 2's pzg_decoder_destroy read a freed context here."""
    exe = _build_cxx("lifetimes")
    ref = os.path.join(ROOT, "tests", "golden", "ref")
    out = subprocess.run([exe, os.path.join(ref, "rfctest2.z"), os.path.join(ref, "rfctest2.gold")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "lifetimes ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_python_pool_outlives_its_context(gpu_ctx, oracle):
    """The Python mirror: a DecoderPool keeps decoding after Context.close(), closes after it, and a dropped pool is freed by
    reference counting alone (no self-referential closure left in the Chunk chain)."""
    import gc
    import weakref
    import pure_zlib_amd as P
    from pure_zlib_amd.incremental import Chunk, DecoderPool, Done
    ctx = P.Context(0)
    pool = DecoderPool(1, ctx)
    z = zlib.compress(corpus.zipf_text(150000, 5), 6)
    ctx.close()
    st, got = pool.start(0).feed(z), b""
    while isinstance(st, Chunk):
        got += st.chunk
        st = st.next()
    assert isinstance(st, Done) and got == zlib.decompress(z)
    with pytest.raises(Exception):
        P.decompress(z, ctx=ctx)  # the closed handle is refused, not dereferenced
    gc.disable()
    try:
        w = weakref.ref(pool)
        st2 = DecoderPool(1, gpu_ctx).start(0).feed(z)  # a chain of Chunks that is simply dropped
        w2 = weakref.ref(st2._rest.args[0])
        del pool, st, st2
        assert w() is None and w2() is None, "a DecoderPool is kept alive by a reference cycle"
    finally:
        gc.enable()


def test_smoke_exits_cleanly_in_a_child_process():
    """smoke() printed its OK line and then died of SIGSEGV at interpreter exit.  Run it as the driver does -- a
    fresh interpreter -- and look at the exit code."""
    import sys
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as e; e.smoke()"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])


def test_parity_of_the_build_without_the_mllvm_options(tmp_path):
    """The speed of libpzg.so depends on two non-default LLVM options (csrc/Makefile KERNELFLAGS); its correctness must not.
    The same sources built without them (through PZG_LIB, in a child process) against the oracle: the nine fixtures, seeded
    valid and corrupt streams on rings 11 and 15, and the incremental protocol's event trace."""
    import sys
    from test_abi import noflags_library
    so = noflags_library()
    code = r'''
import os, sys, zlib
sys.path.insert(0, os.path.join(os.environ["PZG_ROOT"], "tests")); sys.path.insert(0, os.environ["PZG_ROOT"])
import torch; torch.cuda.init()
import corpus
from conftest import REF_CASES, read_case
import pure_zlib_amd as P
from pure_zlib_amd import _ffi
from oracle import oracle as O
assert _ffi.LIB_PATH.endswith("build/noflags/libpzg.so"), _ffi.LIB_PATH
ctx = P.Context(0)
streams, want = [], []
for name in REF_CASES:
    z, g = read_case(name); streams.append(z); want.append(g)
for seed in range(300):
    d = corpus.mixed_data(200 + seed * 211 % 70000, seed) if seed % 3 else corpus.zipf_text(100 + seed * 997 % 90000, seed)
    z = corpus.compress_variant(d, seed) if seed % 2 else zlib.compress(d, 1 + seed % 9)
    if seed % 5 == 0: z = corpus.corrupt(z, seed)
    streams.append(z); want.append(None)
for ring in (11, 15):
    ctx.set_ring_bits(ring)
    got = P.decompress_many(streams, ctx=ctx)
    for z, w, g in zip(streams, want, got):
        r, o = O.decompress(z, 1 << 21)
        if r.status == 0:
            assert g == P.Right(o) and (w is None or o == w), ring
        else:
            assert (not g.is_right()) and g.value.show() == r.message.decode(), (ring, g, r.message)
from pure_zlib_amd.incremental import Chunk, DecoderPool, NeedMore
big = zlib.compress(corpus.zipf_text(150000, 9), 6)
pieces = [big[i:i + 5000] for i in range(0, len(big), 5000)]
pool = DecoderPool(1, ctx); st = pool.start(0); events = [("NeedMore",)]
for p in pieces:
    st = st.feed(p)
    while isinstance(st, Chunk):
        events.append(("Chunk", len(st.chunk))); st = st.next()
    events.append(("NeedMore",) if isinstance(st, NeedMore) else ("Done",))
assert events == O.trace(pieces)[0]
pool.close(); ctx.close()
print("noflags parity ok", len(streams))
'''
    env = dict(os.environ, PZG_LIB=so, PZG_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "noflags parity ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])


def test_host_paths_when_no_thread_can_be_started(tmp_path):
    """the pipelined host paths start helper threads per call (a drainer in the staged / page-locked batch path,
    a drainer and a copier in a large pzg_decoder_feed).  They are joined whatever way the call ends, and when the system has no
    thread to give the issuing thread runs their stages itself.  build/lab_nothreads/libpzg.so is the product with thread creation
    made to fail (tests/tools/lab_build.sh): a staged batch of several ranges, the same batch in page-locked arenas, and a feed of
    600 decoders x 128 KiB (the pipelined path) -- all equal to zlib / the plaintext, in a child process (PZG_LIB)."""
    import sys
    from test_exotic_streams import NOTHREADS_FLAGS, lab_library
    so = lab_library("nothreads", NOTHREADS_FLAGS)
    code = r'''
import os, sys, zlib
sys.path.insert(0, os.path.join(os.environ["PZG_ROOT"], "tests")); sys.path.insert(0, os.environ["PZG_ROOT"])
import numpy as np
import torch; torch.cuda.init()
import corpus
import pure_zlib_amd as P
from pure_zlib_amd import _ffi
from pure_zlib_amd.zlib import PinnedArena
assert _ffi.LIB_PATH.endswith("build/lab_nothreads/libpzg.so"), _ffi.LIB_PATH
ctx = P.Context(0)
texts = [corpus.zipf_text(1024 * (1 + (k * 37) % 48), k) for k in range(128)]
zs = [zlib.compress(t, 6) for t in texts]
n = 9000  # > 96 MiB: several pipelined ranges
pick = np.random.default_rng(3).integers(0, len(zs), size=n)
streams = [zs[k] for k in pick]
streams[11] = streams[11][:-9]
in_len = np.array([len(s) for s in streams], dtype=np.uint64)
out_cap = np.array([len(texts[k]) for k in pick], dtype=np.uint64)
in_off = np.concatenate([[0], np.cumsum((in_len[:-1] + 15) // 16 * 16)]).astype(np.uint64)
out_off = np.concatenate([[0], np.cumsum((out_cap[:-1] + 15) // 16 * 16)]).astype(np.uint64)
h_in = np.zeros(int(in_off[-1] + in_len[-1]) + 16, dtype=np.uint8)
for k, s in enumerate(streams):
    h_in[int(in_off[k]):int(in_off[k]) + len(s)] = np.frombuffer(s, dtype=np.uint8)
for pinned in (False, True):
    if pinned:
        a_in, a_out = PinnedArena(h_in.size), PinnedArena(int(out_off[-1] + out_cap[-1]) + 16)
        a_in.a[:] = h_in; src, dst = a_in.a, a_out.a
    else:
        src, dst = h_in, np.zeros(int(out_off[-1] + out_cap[-1]) + 16, dtype=np.uint8)
    dst[:] = 0xCD
    o_len, o_st, det, used, ad = ctx.decompress_many_raw(src, in_off, in_len, dst, out_off, out_cap, pinned=pinned)
    assert o_st[11] == 1 and int((o_st == 0).sum()) == n - 1, (pinned, int((o_st != 0).sum()))
    for k in range(n):
        if k != 11:
            assert dst[int(out_off[k]):int(out_off[k]) + int(out_cap[k])].tobytes() == texts[pick[k]], (pinned, k)
from pure_zlib_amd import benchmark as HB
plain = [corpus.zipf_text(128 * 1024, 9000 + k) for k in range(8)]
res = HB.incremental_throughput(ctx, [zlib.compress(t, 6) for t in plain], plain, n_decoders=600, passes=1)
assert res["ok"], res
ctx.close()
print("no-thread host paths ok", n, res["feed_calls"])
'''
    env = dict(os.environ, PZG_LIB=so, PZG_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "no-thread host paths ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])


def test_decompress_many_sharded_device_pointers(gpu_ctx, oracle, monkeypatch):
    """pzg_decompress_many_sharded: data ALREADY on the devices -- one batch of device pointers per
    shard, every batch enqueued on its own device's stream by one call, nothing staged through the host.  Three shards
    (the box's own devices when it has several -- every batch then lives on ITS shard's device -- else all on device 0),
    four batches (two on one shard), synchronous and PZG_ASYNC + pzg_sync; every stream against the plaintext, samples
    against the oracle; bad arguments refused."""
    import pure_zlib_amd as P
    import torch
    from devbatch import DeviceBatch
    from pure_zlib_amd._ffi import PzgError
    devs = _shard_devices(3)
    group = P.Context(devices=devs)
    try:
        texts = [corpus.zipf_text(1024 * (1 + (k * 37) % 48), k) for k in range(256)]
        zs = [zlib.compress(t, 6) for t in texts]
        rng = np.random.default_rng(3)
        shard_of = [0, 1, 2, 1]
        parts = [DeviceBatch(texts, zs, rng.integers(0, len(zs), size=n), dev=devs[sh]) for n, sh in zip((5000, 3000, 7000, 11), shard_of)]

        def as_batches():
            return [dict(shard=s, n=b.n, in_base=b.d_in.data_ptr(), in_off=b.d_in_off.data_ptr(), in_len=b.d_in_len.data_ptr(),
                         out_base=b.d_out.data_ptr(), out_off=b.d_out_off.data_ptr(), out_cap=b.d_out_cap.data_ptr(),
                         out_len=b.d_out_len.data_ptr(), status=b.d_status.data_ptr(), detail=b.d_detail.data_ptr(),
                         in_used=b.d_in_used.data_ptr(), adler=b.d_adler.data_ptr()) for s, b in zip(shard_of, parts)]

        def results(b):
            return (b.d_status.cpu().numpy(), b.d_out_len.cpu().numpy(), b.d_in_used.cpu().numpy(), b.d_adler.cpu().numpy().view(np.uint32))
        for sync in (True, False):
            for b in parts:
                b.d_out.fill_(0xCD)
                b.d_status.fill_(-1)
            for d in set(devs):
                torch.cuda.synchronize(d)
            group.decompress_many_sharded(as_batches(), sync=sync, lpt=not sync)
            if not sync:
                group.sync()
            for b in parts:
                b.check_all(*results(b))
            parts[0].check_sample_vs_oracle(oracle, 64)
        bad = as_batches()
        bad[1]["shard"] = 3
        with pytest.raises(PzgError):
            group.decompress_many_sharded(bad)
        bad = as_batches()
        bad[2]["status"] = 0
        with pytest.raises(PzgError):
            group.decompress_many_sharded(bad)
        group.decompress_many_sharded([])
    finally:
        group.close()


def test_host_path_pipelined_ranges_back_to_back_extents(gpu_ctx):
    """The host-pointer path on a batch large enough for several pipelined ranges (192 MiB of output), extents back to back
    and not page aligned: every stream against its plaintext, a truncated stream and one that outgrows its capacity among
    them, and the bytes in front of and behind the extents untouched.  (Round 3 tried downloading such spans straight into
    the caller's arena -- page-locked for the copy -- instead of through pinned staging: measured slower)"""
    texts = [corpus.zipf_text(16384, k) for k in range(512)]
    zs = [zlib.compress(t, 6) for t in texts]
    pick = np.random.default_rng(11).integers(0, len(zs), size=12288)
    streams = [zs[k] for k in pick]
    streams[77] = streams[77][:-9]                      # truncated
    streams[4000] = zlib.compress(texts[3] * 2, 6)      # outgrows its capacity (PZG_E_OUT_TOO_SMALL)
    n = len(streams)
    in_len = np.array([len(s) for s in streams], dtype=np.uint64)
    in_off = np.zeros(n, dtype=np.uint64)
    in_off[1:] = np.cumsum((in_len[:-1] + 15) // 16 * 16)
    in_buf = np.zeros(int(in_off[-1] + in_len[-1]) + 16, dtype=np.uint8)
    for k, s in enumerate(streams):
        in_buf[int(in_off[k]):int(in_off[k]) + len(s)] = np.frombuffer(s, dtype=np.uint8)
    guard = 4096 + 48
    out_cap = np.full(n, 16384, dtype=np.uint64)
    out_off = guard + np.arange(n, dtype=np.uint64) * np.uint64(16384)
    out_buf = np.full(guard + n * 16384 + guard, 0xCD, dtype=np.uint8)
    out_len, status, _detail, _used, _adler = gpu_ctx.decompress_many_raw(in_buf, in_off, in_len, out_buf, out_off, out_cap)
    assert (out_buf[:guard] == 0xCD).all() and (out_buf[-guard:] == 0xCD).all()
    assert status[77] == 1 and status[4000] == 14 and int(out_len[4000]) == 32768
    good = np.ones(n, dtype=bool)
    good[[77, 4000]] = False
    assert (status[good] == 0).all()
    body = out_buf[guard:guard + n * 16384].reshape(n, 16384)
    pool = np.frombuffer(b"".join(texts), dtype=np.uint8).reshape(len(texts), 16384)
    assert np.array_equal(body[good], pool[pick][good])


def test_host_pinned_arenas(gpu_ctx, oracle):
    """PZG_HOST_PINNED: arenas from pzg_host_alloc are read and written by the copy engines directly --
    the path the module mirrors take.  The same batch staged and pinned: identical results; mixed sizes (launched longest
    first through the device-side permutation), a batch big enough for several pipelined ranges, bad streams among them,
    the bytes in front of and behind the spans untouched; extents out of order are refused; a multi-shard context moves
    one span per shard."""
    import pure_zlib_amd as P
    from pure_zlib_amd._ffi import PzgError
    from pure_zlib_amd.zlib import PinnedArena
    texts = [corpus.zipf_text(1024 * (1 + (k * 37) % 48), k) for k in range(256)]
    zs = [zlib.compress(t, 6) for t in texts]
    rng = np.random.default_rng(5)
    for n in (300, 9000):  # one range / several ranges (> 96 MiB)
        pick = rng.integers(0, len(zs), size=n)
        streams = [zs[k] for k in pick]
        streams[7] = streams[7][:-9]                       # truncated
        streams[n - 5] = zlib.compress(texts[3] * 3, 6)    # outgrows its capacity
        in_len = np.array([len(s) for s in streams], dtype=np.uint64)
        out_cap = np.array([len(texts[k]) for k in pick], dtype=np.uint64)
        guard = 4096 + 16
        in_off = guard + np.concatenate([[0], np.cumsum((in_len[:-1] + 15) // 16 * 16)]).astype(np.uint64)
        out_off = guard + np.concatenate([[0], np.cumsum((out_cap[:-1] + 15) // 16 * 16)]).astype(np.uint64)
        a_in = PinnedArena(int(in_off[-1] + in_len[-1]) + guard)
        a_out = PinnedArena(int(out_off[-1] + out_cap[-1]) + guard)
        a_in.a[:] = 0xEE
        for k, s in enumerate(streams):
            a_in.a[int(in_off[k]):int(in_off[k]) + len(s)] = np.frombuffer(s, dtype=np.uint8)
        ref_out = np.full(a_out.nbytes, 0xCD, dtype=np.uint8)
        ref = gpu_ctx.decompress_many_raw(a_in.a.copy(), in_off, in_len, ref_out, out_off, out_cap)  # staged, pageable
        for ctx in (gpu_ctx, None):
            group = None
            if ctx is None:
                group = ctx = P.Context(devices=_shard_devices(3))
            try:
                a_out.a[:] = 0xCD
                got = ctx.decompress_many_raw(a_in.a, in_off, in_len, a_out.a, out_off, out_cap, pinned=True)
                for x, y in zip(ref, got):
                    assert np.array_equal(x, y)
                assert got[1][7] == 1 and got[1][n - 5] == 14 and int((got[1] == 0).sum()) == n - 2
                for k in range(n):  # (a failed stream's bytes are unspecified: only what it produced before the error was flushed)
                    if got[1][k] not in (0, 14):
                        continue
                    nb = int(min(got[0][k], out_cap[k]))
                    lo = int(out_off[k])
                    assert np.array_equal(a_out.a[lo:lo + nb], ref_out[lo:lo + nb]), k
                assert (a_out.a[:guard] == 0xCD).all() and (a_out.a[int(out_off[-1] + out_cap[-1]):] == 0xCD).all()
            finally:
                if group is not None:
                    group.close()
        r, o = oracle.decompress(streams[11], int(out_cap[11]))
        assert r.status == 0 and a_out.a[int(out_off[11]):int(out_off[11]) + int(out_cap[11])].tobytes() == o
        # extents must ascend
        swapped = out_off.copy()
        swapped[[3, 4]] = swapped[[4, 3]]
        with pytest.raises(PzgError):
            gpu_ctx.decompress_many_raw(a_in.a, in_off, in_len, a_out.a, swapped, out_cap, pinned=True)
        a_in.close()
        a_out.close()
    # the mirrors pack into page-locked arenas themselves
    assert P.decompress_many(zs[:40], ctx=gpu_ctx) == [P.Right(t) for t in texts[:40]]


def test_overlapping_async_launches_on_two_streams(gpu_ctx):
    """two PZG_ASYNC | PZG_LPT_ORDER launches (one of them gzip) enqueued on two different user streams
    may run at the same time: each has its own work counter, launch permutation and expected-CRC array."""
    import struct
    import torch
    from devbatch import DeviceBatch
    texts = [corpus.zipf_text(1024 * (1 + (s * 2654435761 >> 7) % 64), s) for s in range(128)]
    zs = [zlib.compress(t, 6) for t in texts]
    gz = [b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + z[2:-4] + struct.pack("<II", zlib.crc32(t), len(t)) for z, t in zip(zs, texts)]
    rng = np.random.default_rng(21)
    a = DeviceBatch(texts, zs, rng.integers(0, len(zs), size=30000))
    b = DeviceBatch(texts, gz, rng.integers(0, len(zs), size=20000))
    c = DeviceBatch(texts, zs, rng.integers(0, len(zs), size=25000))
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    try:
        for rounds in range(3):
            for x in (a, b, c):
                x.d_out.fill_(0xCD)
                x.d_status.fill_(-1)
            torch.cuda.synchronize()
            for x, st, g in ((a, s1, False), (b, s2, True), (c, s3, False)):
                gpu_ctx.set_stream(st.cuda_stream)
                gpu_ctx.decompress_many_device(x.d_in.data_ptr(), x.d_in_off.data_ptr(), x.d_in_len.data_ptr(), x.d_out.data_ptr(),
                                               x.d_out_off.data_ptr(), x.d_out_cap.data_ptr(), x.d_out_len.data_ptr(), x.d_status.data_ptr(),
                                               x.d_detail.data_ptr(), x.d_in_used.data_ptr(), x.d_adler.data_ptr(), x.n, sync=False, gzip=g, lpt=True)
            torch.cuda.synchronize()
            a.check_all(a.d_status.cpu().numpy(), a.d_out_len.cpu().numpy(), a.d_in_used.cpu().numpy(), a.d_adler.cpu().numpy().view(np.uint32))
            c.check_all(c.d_status.cpu().numpy(), c.d_out_len.cpu().numpy(), c.d_in_used.cpu().numpy(), c.d_adler.cpu().numpy().view(np.uint32))
            assert (b.d_status.cpu().numpy() == 0).all() and (b.d_out_len.cpu().numpy() == b.out_cap).all()
            crc = np.array([zlib.crc32(t) for t in texts], dtype=np.uint32)[b.pick]
            assert (b.d_adler.cpu().numpy().view(np.uint32) == crc).all()
    finally:
        gpu_ctx.reset_stream()
