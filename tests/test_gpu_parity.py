"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the
reference's own golden vectors.  Bit-exact (integer/byte work): no tolerance anywhere."""
import zlib

import numpy as np
import pytest

import corpus
from conftest import REF_CASES, read_case

pytestmark = pytest.mark.gpu


def run_batch(ctx, streams, caps, align=16):
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
    res = ctx.decompress_many_raw(in_buf, in_off, in_len, out_buf, out_off, out_cap)
    outs = [out_buf[int(out_off[k]):int(out_off[k]) + min(int(res[0][k]), caps[k])].tobytes() for k in range(n)]
    return res, outs, out_buf, out_off


@pytest.mark.parametrize("name", REF_CASES)
def test_reference_golden_decompress(gpu_ctx, name):
    """test/Test.hs:83-86: assertEqual (Right gold) (decompress z), through the mirror API."""
    import pure_zlib_amd as P
    z, gold = read_case(name)
    assert P.decompress(z, ctx=gpu_ctx) == P.Right(gold)


def test_reference_golden_batch(gpu_ctx, oracle):
    zs, golds = zip(*[read_case(n) for n in REF_CASES])
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(gpu_ctx, list(zs), [len(g) for g in golds])
    for k, name in enumerate(REF_CASES):
        assert status[k] == 0, name
        assert outs[k] == golds[k], name
        assert int(out_len[k]) == len(golds[k])
        assert int(in_used[k]) == len(zs[k])
        assert int(adler[k]) == zlib.adler32(golds[k])


def test_valid_streams_vs_oracle(gpu_ctx, oracle):
    streams, datas = [], []
    for seed in range(600):
        n = [0, 1, 2, 5, 100, 1000, 5000, 40000, 70000, 200000][seed % 10] if seed % 7 == 0 else (seed * 37) % 20000
        d = corpus.mixed_data(n, seed)
        streams.append(corpus.compress_variant(d, seed))
        datas.append(d)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(gpu_ctx, streams, [len(d) for d in datas])
    for k in range(len(streams)):
        r, o = oracle.decompress(streams[k], len(datas[k]))
        assert r.status == 0 and o == datas[k]
        assert status[k] == 0, (k, status[k], detail[k])
        assert outs[k] == datas[k], k
        assert int(adler[k]) == r.adler and int(in_used[k]) == r.in_used and int(out_len[k]) == r.out_len


def test_corrupt_streams_vs_oracle(gpu_ctx, oracle):
    streams, caps = [], []
    for seed in range(3000):
        d = corpus.mixed_data((seed * 131) % 3000 + 1, seed)
        z = corpus.corrupt(corpus.compress_variant(d, seed), seed)
        streams.append(z)
        caps.append([len(d), len(d) + 100, 1 << 17][seed % 3])
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(gpu_ctx, streams, caps)
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


def test_output_never_written_past_capacity(gpu_ctx):
    d = corpus.zipf_text(50000, 3)
    z = zlib.compress(d, 6)
    caps = [0, 1, 15, 16, 17, 4095, 4096, 32768, 49999]
    (out_len, status, detail, in_used, adler), outs, out_buf, out_off = run_batch(gpu_ctx, [z] * len(caps), caps)
    for k, cap in enumerate(caps):
        assert status[k] == 14 and int(out_len[k]) == len(d)
        assert outs[k] == d[:cap]
        lo = int(out_off[k]) + cap
        hi = int(out_off[k + 1]) if k + 1 < len(caps) else lo
        assert (out_buf[lo:hi] == 0xCD).all()  # padding between extents untouched


def test_unaligned_extents(gpu_ctx):
    streams, datas = [], []
    for seed in range(64):
        d = corpus.mixed_data(1000 + seed * 97, seed + 11)
        datas.append(d)
        streams.append(zlib.compress(d, 1 + seed % 9))
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(gpu_ctx, streams, [len(d) for d in datas], align=1)
    for k in range(len(streams)):
        assert status[k] == 0 and outs[k] == datas[k], k


def test_adler32_kernel(gpu_ctx, oracle):
    rng = np.random.default_rng(5)
    for n in [0, 1, 15, 16, 17, 1000, 65535, 65536, 65537, 1 << 20, (1 << 22) + 12345, 50_000_000]:
        buf = rng.integers(0, 256, size=n + 32, dtype=np.uint8)
        for skew in (0, 3):
            view = buf[skew:skew + n]
            got = gpu_ctx.adler32(view)
            assert got == zlib.adler32(view.tobytes()), (n, skew)
    worst = np.full(3_000_000, 255, dtype=np.uint8)
    assert gpu_ctx.adler32(worst) == zlib.adler32(worst.tobytes())
    assert gpu_ctx.adler32(worst[:70000], init=0xFFF0FFF0 % (1 << 32)) == zlib.adler32(worst[:70000].tobytes(), 0xFFF0FFF0)
