"""GPU: the bundles (pzg_bundle_kernel.h, bundle_core.h -- a device-pointer launch takes its streams of the fixed code 64 to a
wavefront, one lane per stream) against the oracle, through the C ABI.  Reference semantics:
Deflate.hs:79-82 (BTYPE 1), 106-120 (runInflate), 241-251 (the fixed trees); everything a bundle does not take is the ordinary
kernel's, and nothing about a result may depend on which of the two produced it."""
import random
import zlib

import numpy as np
import pytest

import corpus
from devbatch import DeviceBatch
from test_model_bundles import fixed

pytestmark = pytest.mark.gpu


def _status_map(b, oracle):
    """expected (status, out_len, in_used, adler, bytes) per pool entry, from the oracle"""
    exp = []
    for t, z in zip(b.texts, b.zs):
        r, o = oracle.decompress(z, len(t))
        exp.append((r, o))
    return exp


def _check_against_oracle(b, res, oracle):
    status, out_len, in_used, adler = res
    exp = _status_map(b, oracle)
    detail = b.d_detail.cpu().numpy().view(np.uint32)
    out = b.d_out.cpu().numpy()
    for k in range(b.n):
        r, o = exp[b.pick[k]]
        assert status[k] == r.status, (k, b.pick[k], status[k], r.status)
        if r.status == 0:
            lo = int(b.out_off[k])
            assert out_len[k] == r.out_len and in_used[k] == r.in_used and adler[k] == r.adler, (k, b.pick[k])
            assert out[lo:lo + len(o)].tobytes() == o, (k, b.pick[k])
        elif r.status in (3, 4, 6, 10, 11, 12, 13):
            assert (int(detail[2 * k]), int(detail[2 * k + 1])) == (r.detail0, r.detail1), (k, b.pick[k], r.status)


def _mixed_pool(rng):
    """plain streams of the fixed code of every small size and kind, and everything that is NOT a bundle's business"""
    texts, zs = [], []

    def add(t, z):
        texts.append(t)
        zs.append(z)
    for k in range(160):
        kind = k % 5
        n = rng.randrange(0, 4097)
        if kind == 0:
            d = corpus.zipf_text(n, k)
        elif kind == 1:
            d = bytes(b % 144 for b in corpus.random_bytes(n, k))
        elif kind == 2:
            d = bytes([65 + k % 7]) * n
        elif kind == 3:
            d = (corpus.zipf_text(300, k) * 20)[:n]
        else:
            d = corpus.mixed_data(n, k)
        add(d, fixed(d, level=rng.choice([1, 6, 9]), blocks=rng.choice([1, 1, 2, 5, 17])))
    d = corpus.zipf_text(3000, 1)
    good = fixed(d)
    add(d, zlib.compress(d, 6))
    add(corpus.random_bytes(500, 1), zlib.compress(corpus.random_bytes(500, 1), 0))
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    add(d, co.compress(d[:1000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d[1000:]) + co.flush())
    for cut in (1, 5, len(good) // 2, len(good) - 2):
        add(d, good[:-cut])
    add(d[:-1], good)      # capacity one byte short
    add(b"", good)         # no capacity at all
    add(d, good[:-4] + bytes([good[-4] ^ 1]) + good[-3:])
    add(d, bytes([0x78, 0x9d]) + good[2:])
    z = bytearray(good)
    z[1] |= 0x20
    add(d, bytes(z))
    big = corpus.zipf_text(20000, 3)
    add(big, fixed(big))
    add(big, zlib.compress(big, 6))
    for k in range(60):
        z = bytearray(good)
        p = rng.randrange(2, len(z) - 4)
        z[p] ^= 1 << rng.randrange(8)
        add(d, bytes(z))
    return texts, zs


@pytest.mark.parametrize("ring", [11, 15])
def test_mixed_batch_with_and_without_bundles(gpu_ctx, oracle, ring):
    rng = random.Random(0xB0)
    texts, zs = _mixed_pool(rng)
    pick = np.random.default_rng(0xB1).integers(0, len(zs), size=8192)
    pick[: len(zs)] = np.arange(len(zs))  # every pool entry at least once
    b = DeviceBatch(texts, zs, pick)
    try:
        for on in (2, 0, 2):  # (2: bundles whatever the batch's size -- the default takes 32,768 streams or more)
            gpu_ctx.set_bundles(on)
            res = b.run(gpu_ctx, ring)
            _check_against_oracle(b, res, oracle)
    finally:
        gpu_ctx.set_bundles(1)
        gpu_ctx.set_ring_bits(11)


def test_fewer_streams_than_a_bundle_and_odd_counts(gpu_ctx, oracle):
    texts = [corpus.zipf_text(100 + 37 * k, k) for k in range(100)]
    zs = [fixed(t) for t in texts]
    try:
        gpu_ctx.set_bundles(2)
        for n in (1, 63, 64, 65, 100):
            b = DeviceBatch(texts, zs, np.arange(n))
            res = b.run(gpu_ctx, 11)
            b.check_all(*res)
    finally:
        gpu_ctx.set_bundles(1)


def test_large_streams_of_the_fixed_code_in_bundles(gpu_ctx, oracle):
    """Far matches (older than a lane's 512-byte window: read from the stream's own flushed output), long matches, matches that
    overlap themselves, streams of very different lengths in one bundle."""
    texts = [corpus.zipf_text(30000 + 1111 * k, k) for k in range(20)] + [bytes([7]) * 70000, (corpus.zipf_text(700, 9) * 90)[:60000],
             bytes(b % 144 for b in corpus.random_bytes(40000, 5)), corpus.mixed_data(50000, 2), corpus.mixed_data(50000, 3)]
    texts += [corpus.zipf_text(10 + 3 * k, k) for k in range(40)]
    zs = [fixed(t, level=9 if k % 2 else 1, blocks=1 + k % 3) for k, t in enumerate(texts)]
    pick = np.random.default_rng(0xB3).permutation(np.arange(4 * len(zs)) % len(zs))
    b = DeviceBatch(texts, zs, pick)
    try:
        gpu_ctx.set_bundles(2)
        res = b.run(gpu_ctx, 11)
        b.check_all(*res)
        b.check_sample_vs_oracle(oracle, 64)
    finally:
        gpu_ctx.set_bundles(1)


def test_config3_share_at_the_default_setting(gpu_ctx, oracle):
    """40,000 streams of 4 KiB (the default takes launches of 32,768 streams or more), every byte, and the same launch with the bundles off."""
    texts = [corpus.zipf_text(4096, s) for s in range(256)]
    zs = [fixed(t) for t in texts]
    pick = np.random.default_rng(0xB2).integers(0, len(zs), size=40000)
    b = DeviceBatch(texts, zs, pick)
    try:
        for on in (1, 0):
            gpu_ctx.set_bundles(on)
            res = b.run(gpu_ctx, 11)
            b.check_all(*res)
            b.check_sample_vs_oracle(oracle, 64)
    finally:
        gpu_ctx.set_bundles(1)


def test_setting_the_option_makes_the_next_launch_look_again(gpu_ctx):
    """A context whose launches had nothing for the bundles goes 2, 6, 14 ... 64 launches without looking (pzg.h PZG_OPT_BUNDLES); setting
    the option -- to the value it has -- ends that.  Four launches of dynamic-code streams, then streams of the fixed code: decoded by
    the ordinary kernel (the context is not looking), and after set_bundles(1) by the bundles -- told apart by the kernels' time (config
    3's streams: 1.4 ms against 1.1), every byte checked both ways."""
    texts = [corpus.zipf_text(4096, s) for s in range(256)]
    dyn = DeviceBatch(texts, [zlib.compress(t, 6) for t in texts], np.random.default_rng(1).integers(0, 256, size=40000))
    fx = DeviceBatch(texts, [fixed(t) for t in texts], np.random.default_rng(2).integers(0, 256, size=65536))
    try:
        gpu_ctx.set_bundles(1)
        for _ in range(4):
            dyn.check_all(*dyn.run(gpu_ctx, 11))
        slow = []
        for _ in range(2):
            fx.check_all(*fx.run(gpu_ctx, 11))
            slow.append(gpu_ctx.last_kernel_ms())
        gpu_ctx.set_bundles(1)
        fast = []
        for _ in range(3):
            fx.check_all(*fx.run(gpu_ctx, 11))
            fast.append(gpu_ctx.last_kernel_ms())
        assert min(fast[1:]) < 0.92 * min(slow), (slow, fast)
    finally:
        gpu_ctx.set_bundles(1)
