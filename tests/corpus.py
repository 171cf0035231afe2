"""Seeded synthetic corpora shared by the tests and bench.py.

All compression is done with the system zlib (Python's `zlib`), which SURVEY.md section 0 item 3
establishes as an independent, specification-equal encoder/decoder for valid streams.
"""
import random
import zlib

import numpy as np

_ALPHA = b"abcdefghijklmnopqrstuvwxyz"


_VOCAB_CACHE = {}


def _vocab(vocab: int, s: float):
    """Seed-independent vocabulary (fixed seed 0x5EED): word matrix [vocab, 11] (trailing sep slot), lengths, Zipf CDF."""
    key = (vocab, s)
    if key not in _VOCAB_CACHE:
        rng = np.random.default_rng(0x5EED)
        lens = rng.integers(2, 11, size=vocab)
        mat = np.frombuffer(_ALPHA, dtype=np.uint8)[rng.integers(0, 26, size=(vocab, 11))].copy()
        ranks = np.arange(1, vocab + 1, dtype=np.float64)
        prob = ranks ** (-s)
        cdf = np.cumsum(prob / prob.sum())
        _VOCAB_CACHE[key] = (mat, lens.astype(np.int64), cdf)
    return _VOCAB_CACHE[key]


def zipf_text(nbytes: int, seed: int, vocab: int = 4096, s: float = 1.1) -> bytes:
    """Text-like data: words of a fixed 4096-word vocabulary drawn Zipf(1.1) with a
    per-blob seed, space-joined, newline every 256 words.  Vectorised: ~1 ms per 32 KiB."""
    if nbytes == 0:
        return b""
    mat, wl, cdf = _vocab(vocab, s)
    rng = np.random.default_rng(seed)
    nw = nbytes // 3 + 8  # shortest word + separator is 3 bytes
    idx = np.searchsorted(cdf, rng.random(nw)).clip(0, vocab - 1)
    lens = wl[idx] + 1  # + separator
    ends = np.cumsum(lens)
    starts = ends - lens
    total = int(ends[-1])
    word_of = np.repeat(np.arange(nw), lens)
    ch = np.arange(total) - starts[word_of]
    out = mat[idx[word_of], np.minimum(ch, 10)]
    sep_pos = ends - 1
    out[sep_pos] = 0x20
    out[sep_pos[255::256]] = 0x0A
    return out[:nbytes].tobytes()


def random_bytes(nbytes: int, seed: int) -> bytes:
    return np.random.default_rng(seed).integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()


def mixed_data(nbytes: int, seed: int) -> bytes:
    """One of: random, text, a run of one byte, a short repeating pattern with noise."""
    kind = seed % 4
    if kind == 0:
        return random_bytes(nbytes, seed)
    if kind == 1:
        return zipf_text(nbytes, seed, vocab=256)
    if kind == 2:
        return bytes([seed % 251]) * nbytes
    rng = random.Random(seed)
    unit = b"ab" * 7 + bytes(rng.getrandbits(8) for _ in range(5))
    return (unit * (nbytes // len(unit) + 1))[:nbytes]


def compress_variant(data: bytes, seed: int) -> bytes:
    """zlib stream with seeded level/strategy/window/memLevel and seeded flush points (multi-block)."""
    rng = random.Random(seed)
    lvl = rng.randint(0, 9)
    strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
    co = zlib.compressobj(lvl, zlib.DEFLATED, rng.randint(9, 15), rng.randint(1, 9), strat)
    z, pos = b"", 0
    while pos < len(data):
        step = rng.randint(1, max(1, len(data)))
        z += co.compress(data[pos:pos + step])
        pos += step
        if rng.random() < 0.3:
            z += co.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
    return z + co.flush()


def corrupt(z: bytes, seed: int) -> bytes:
    rng = random.Random(seed)
    b = bytearray(z)
    mode = rng.randrange(3)
    if mode == 0 and b:
        for _ in range(rng.randint(1, 3)):
            i = rng.randrange(len(b))
            b[i] ^= 1 << rng.randrange(8)
    elif mode == 1 and b:
        b = b[:rng.randrange(len(b))]
    elif b:
        i = rng.randrange(len(b))
        b[i:i + rng.randint(1, 4)] = bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 4)))
    return bytes(b)


def skewed_bytes(nbytes: int, seed: int) -> bytes:
    """Literal-heavy binary-like data: all 256 byte values under a geometric law, so a dynamic code has
    many literals longer than the primary table (the kernel's second-level tables) and few matches."""
    r = np.random.default_rng(seed)
    t = np.minimum(r.geometric(0.03, size=nbytes) - 1, 255).astype(np.uint16)
    return ((t * 151 + 7) % 256).astype(np.uint8).tobytes()


_HTML = None


def binary_records(nbytes: int, seed: int) -> bytes:
    """Binary-looking data: 16-byte records -- a little-endian counter, a small enum, two random bytes, constants.  Its literal/length
    code has ~250 symbols of 8 to 10 bits: more second-level entries than the kernels' pool holds (bench.py's hetero workload, kind 3)."""
    rng = np.random.default_rng(0xB1A0 + seed)
    nrec = nbytes // 16
    rec = np.zeros((nrec, 16), dtype=np.uint8)
    rec[:, 0:4] = (np.arange(nrec, dtype=np.uint32) * 3 + seed).view(np.uint8).reshape(nrec, 4)
    rec[:, 4] = rng.integers(0, 4, size=nrec)
    rec[:, 8:10] = rng.integers(0, 256, size=(nrec, 2))
    rec[:, 12] = 0xff
    rec[:, 13] = rng.integers(0, 2, size=nrec) * 0x80
    return rec.tobytes()


def html_slice(nbytes: int, seed: int) -> bytes:
    """A slice of the reference's own RFC html fixtures (tests/golden/ref/rfctest*.gold): ~95 distinct
    symbols, literal/length codes up to 13 bits."""
    global _HTML
    if _HTML is None:
        import os
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref")
        _HTML = b"".join(open(os.path.join(d, f"rfctest{i}.gold"), "rb").read() for i in (1, 2, 3))
    o = (seed * 7919) % max(1, len(_HTML) - nbytes)
    return _HTML[o:o + nbytes]


def fixed_blob(nbytes: int, seed: int) -> bytes:
    """BASELINE config 3: level-1 with Z_FIXED so the block really is BTYPE=1."""
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    return co.compress(zipf_text(nbytes, seed)) + co.flush()


def level6_blob(nbytes: int, seed: int) -> bytes:
    """BASELINE config 4/5: zlib.compress(text, 6): one dynamic block at these sizes."""
    return zlib.compress(zipf_text(nbytes, seed), 6)


def gzip_member(data: bytes, seed: int) -> bytes:
    """One RFC 1952 member with seeded level and seeded optional header fields (FEXTRA, FNAME, FCOMMENT,
    FHCRC), assembled by hand around a raw deflate body so every header form is exercised."""
    import struct
    rng = random.Random(seed)
    co = zlib.compressobj(1 + seed % 9, zlib.DEFLATED, -15, rng.randint(1, 9),
                          rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_FILTERED]))
    body = co.compress(data) + co.flush()
    flg = rng.randrange(32) & 0x1e
    hdr = bytearray(b"\x1f\x8b\x08" + bytes([flg]) + struct.pack("<I", (seed * 977) & 0xffffffff) + bytes([rng.choice([0, 2, 4]), rng.choice([0, 3, 255])]))
    if flg & 4:
        extra = bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 40)))
        hdr += struct.pack("<H", len(extra)) + extra
    if flg & 8:
        hdr += bytes(rng.randint(1, 255) for _ in range(rng.randint(0, 30))) + b"\x00"
    if flg & 16:
        hdr += bytes(rng.randint(1, 255) for _ in range(rng.randint(0, 300))) + b"\x00"
    if flg & 2:
        hdr += struct.pack("<H", zlib.crc32(bytes(hdr)) & 0xffff)
    return bytes(hdr) + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


# ---- BASELINE config 2's buffer: byte i = byte (i & 7), little-endian, of splitmix64(seed + (i >> 3)) -----------
SPLITMIX_SEED = 0x5EED0002


def splitmix64_numpy(first, count):
    """splitmix64(SPLITMIX_SEED + j) for j = first .. first + count - 1, as uint64."""
    with np.errstate(over="ignore"):
        z = (np.arange(first, first + count, dtype=np.uint64) + np.uint64(SPLITMIX_SEED)) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def splitmix64_torch(first, count, dev):
    """The same on the device: int64 arithmetic wraps like uint64; the logical right shifts are masked arithmetic ones."""
    import torch
    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def i64(c):
        return c - (1 << 64) if c >= (1 << 63) else c
    z = torch.arange(first, first + count, dtype=torch.int64, device=dev) + (SPLITMIX_SEED + i64(0x9E3779B97F4A7C15))
    z = (z ^ lsr(z, 30)) * i64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def strip_case(seed: int):
    """(data, stream) long enough for the kernels' strips (spans of 64 x 256 input bits or more: the shortest span until round 6; 64 x 64 bits since): text, html, literal-heavy
    bytes, four-symbol data (two-bit codes: the shortest strips), pieces of all of them in one stream; one block or many
    (seeded flush points, fixed / Huffman-only / RLE strategies, stored blocks at level 0)."""
    rng = random.Random(90000 + seed)
    n = rng.choice([24000, 40000, 70000, 150000])
    kind = seed % 6
    if kind == 0:
        d = zipf_text(n, seed)
    elif kind == 1:
        d = html_slice(min(n, 60000), seed)
    elif kind == 2:
        d = skewed_bytes(n, seed)
    elif kind == 3:
        d = bytes(rng.choice(b"acgt") for _ in range(n))
    elif kind == 4:
        d = zipf_text(n // 3, seed) + skewed_bytes(n // 3, seed) + bytes([seed % 251]) * 5000 + zipf_text(n // 3, seed + 1)
    else:
        d = mixed_data(n, seed)
    if seed % 3 == 0:
        z = zlib.compress(d, 1 + seed % 9)
    else:
        z = compress_variant(d, seed)
    return d, z
