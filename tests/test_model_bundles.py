"""CPU: the bundles (inflate_core.h bundle_decode / bundle_emit -- small streams of the fixed code, 64 to a wave, one lane per
stream) compiled as a host program and checked against the oracle: what a bundle calls clean must be the oracle's result bit for
bit, everything else must come back as "the ordinary kernel's" (status 103) -- and plain streams of the fixed code must be clean.
The model is test infrastructure; libpzg.so never contains it.  Reference semantics: Deflate.hs:79-82, 106-120, 241-251."""
import ctypes as C
import random
import zlib

import pytest

import corpus
from oracle import oracle as O
from test_model_vs_oracle import R, _build_model, same  # noqa: F401  (builds tests/model/libpzgmodel.so)

TODO = 103


@pytest.fixture(scope="session")
def bundle():
    import os
    from conftest import ROOT
    _build_model([])
    M = C.CDLL(os.path.join(ROOT, "tests", "model", "libpzgmodel.so"))

    def run(streams, caps):
        n = len(streams)
        assert n <= 64
        ins = (C.c_char_p * n)(*streams)
        lens = (C.c_uint64 * n)(*[len(z) for z in streams])
        bufs = [C.create_string_buffer(max(c, 1) + 64) for c in caps]
        outs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        capa = (C.c_uint64 * n)(*caps)
        res = (R * n)()
        M.pzm_bundle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        assert M.pzm_bundle(ins, lens, outs, capa, n, res) == 0
        return [(res[k], bufs[k].raw[: min(res[k].out_len, caps[k])]) for k in range(n)]
    return run


def fixed(data, level=1, blocks=1):
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    if blocks == 1:
        return co.compress(data) + co.flush()
    step = max(1, len(data) // blocks)
    z = b""
    for i in range(0, len(data), step):
        z += co.compress(data[i:i + step]) + co.flush(zlib.Z_BLOCK)  # ends the block; the next one is of the fixed code again
    return z + co.flush()


def check(bundle, streams, caps, expect_clean=None):
    got = bundle(streams, caps)
    nclean = 0
    for k, (z, cap, (r, out)) in enumerate(zip(streams, caps, got)):
        if r.status == TODO:
            assert expect_clean is None or not expect_clean[k], (k, len(z), cap)
            continue
        nclean += 1
        ro, oo = O.decompress(z, cap)
        assert same(ro, oo, r, out), (k, ro.status, r.status, ro.out_len, r.out_len, ro.in_used, r.in_used, hex(ro.adler), hex(r.adler))
        assert expect_clean is None or expect_clean[k], (k, "a stream that is not a bundle's came back clean", r.status)
    return nclean


def test_bundle_of_text_streams_of_every_size(bundle):
    datas = [corpus.zipf_text(n, 100 + n) for n in [0, 1, 2, 3, 7, 64, 255, 256, 257, 1000, 2048, 3000, 4095, 4096] + [random.Random(5).randrange(1, 4097) for _ in range(50)]]
    streams = [fixed(d) for d in datas]
    n = check(bundle, streams, [len(d) for d in datas], [True] * len(datas))
    assert n == len(datas)


def test_literal_heavy_runs_and_many_blocks(bundle):
    rng = random.Random(7)
    datas, streams = [], []
    for k in range(64):
        kind = k % 4
        if kind == 0:
            d = bytes(b % 144 for b in corpus.random_bytes(rng.randrange(1, 4096), k))  # nothing but (8-bit) literals: runs of more than 255 of them
        elif kind == 1:
            d = bytes([65 + k % 7]) * rng.randrange(1, 4096)            # distance 1, length 258
        elif kind == 2:
            d = (corpus.zipf_text(300, k) * 20)[: rng.randrange(1, 4096)]  # long matches at a distance of 300
        else:
            d = corpus.zipf_text(rng.randrange(100, 4096), k)
        datas.append(d)
        streams.append(fixed(d, level=rng.choice([1, 6, 9]), blocks=rng.choice([1, 1, 2, 5, 17])))
    assert check(bundle, streams, [len(d) for d in datas], [True] * 64) == 64


def test_what_is_not_a_bundles_business_is_left_alone(bundle):
    """Dynamic and stored blocks, every kind of damage, capacities too small: the bundle must not call them clean -- except a wrong
    checksum, which it reports as the ordinary path does (detail words and all)."""
    rng = random.Random(11)
    streams, caps, exp = [], [], []

    def add(z, cap, clean):
        streams.append(z)
        caps.append(cap)
        exp.append(clean)
    d = corpus.zipf_text(3000, 1)
    good = fixed(d)
    add(good, len(d), True)
    add(zlib.compress(d, 6), len(d), False)                        # a dynamic block
    add(zlib.compress(corpus.random_bytes(500, 1), 0), 500, False)  # a stored block
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    add(co.compress(d[:1000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(d[1000:]) + co.flush(), len(d), False)  # fixed, stored (empty), fixed
    add(good[:-1], len(d), False)                                  # the trailer is short
    add(good[:-5], len(d), False)
    add(good[: len(good) // 2], len(d), False)                     # truncated inside the tokens
    add(good, len(d) - 1, False)                                   # capacity too small
    add(good, 0, False)
    add(good[:-4] + bytes([good[-4] ^ 1]) + good[-3:], len(d), True)  # a wrong checksum: reported with both words
    add(bytes([0x78, 0x9d]) + good[2:], len(d), False)             # FCHECK
    add(bytes([0x79, 0x9c - 0x1f + 0x1f]) + good[2:], len(d), False)
    z = bytearray(good)
    z[1] |= 0x20                                                   # FDICT (and FCHECK now wrong)
    add(bytes(z), len(d), False)
    add(good + b"trailing", len(d), True)                          # bytes behind the trailer are not the stream's (Zlib.hs:46-49 is the mirror's business)
    add(b"", 10, False)
    add(good[:2], 10, False)
    add(good[:7], 10, False)
    big = fixed(corpus.zipf_text(20000, 3))
    add(big, 20000, True)                                          # distances of up to 20,000: far matches all the way
    for k in range(40):                                            # flipped bits anywhere: bad distances, bad symbols, wrong lengths ...
        z = bytearray(good)
        p = rng.randrange(2, len(z) - 4)
        z[p] ^= 1 << rng.randrange(8)
        add(bytes(z), len(d), None)
    got = bundle(streams, caps)
    for k, (z, cap, e, (r, out)) in enumerate(zip(streams, caps, exp, got)):
        if r.status == TODO:
            assert e is not True, k
            continue
        assert e is not False, (k, r.status)
        ro, oo = O.decompress(z, cap)
        assert same(ro, oo, r, out), (k, ro.status, r.status)
        if e is None:
            assert ro.status in (0, 10), (k, ro.status)  # a flipped bit the bundle decoded through: valid, or only the checksum differs


def test_fewer_than_64_streams_and_large_ones(bundle):
    datas = [corpus.zipf_text(1500 + 100 * k, k) for k in range(5)]
    streams = [fixed(d) for d in datas]
    assert check(bundle, streams, [len(d) for d in datas], [True] * 5) == 5
    # larger streams: far matches (older than the lane's 512-byte window), long matches, matches that overlap themselves, a level-9 parse
    datas = [corpus.zipf_text(30000 + 1111 * k, k) for k in range(6)] + [bytes([7]) * 70000, (corpus.zipf_text(700, 9) * 90)[:60000],
             bytes(b % 144 for b in corpus.random_bytes(40000, 5)), corpus.mixed_data(50000, 2), corpus.mixed_data(50000, 3)]
    streams = [fixed(d, level=9 if k % 2 else 1, blocks=1 + k % 3) for k, d in enumerate(datas)]
    assert check(bundle, streams, [len(d) for d in datas], [True] * len(datas)) == len(datas)
