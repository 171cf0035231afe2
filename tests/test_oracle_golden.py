"""CPU: the oracle (oracle/pz_oracle.c) pinned against every vector the reference's own tests hold
for this path, cross-checked against system zlib, and against the pinned generated vectors."""
import hashlib
import json
import os
import zlib

import pytest

import corpus
from conftest import REF_CASES, ROOT, read_case


@pytest.mark.parametrize("name", REF_CASES)
def test_reference_gold_files(oracle, name):
    """test/Test.hs:83-86  assertEqual (Right gold) (decompress z)."""
    z, gold = read_case(name)
    r, out = oracle.decompress(z, len(gold) + 16)
    assert r.status == oracle.OK and r.message == b""
    assert out == gold
    assert r.in_used == len(z) and r.adler == zlib.adler32(gold) and r.quirks == 0


def test_kat_rfc1951_code_generation(oracle):
    """test/Test.hs:13-35,107-113: RFC 1951 3.2.2 example."""
    lengths = [(ord(c), l) for c, l in zip("ABCDEFGH", [3, 3, 3, 3, 3, 2, 4, 4])]
    expect = [(ord("A"), 3, 2), (ord("B"), 3, 3), (ord("C"), 3, 4), (ord("D"), 3, 5), (ord("E"), 3, 6),
              (ord("F"), 2, 0), (ord("G"), 4, 14), (ord("H"), 4, 15)]
    assert oracle.compute_code_values(lengths) == expect


def test_kat_fixed_huffman_lengths(oracle):
    """test/Test.hs:37-52,114-120: the fixed literal/length code."""
    lengths = ([(x, 8) for x in range(144)] + [(x, 9) for x in range(144, 256)] +
               [(x, 7) for x in range(256, 280)] + [(x, 8) for x in range(280, 288)])
    expect = ([(a, 8, b) for a, b in zip(range(144), range(48, 192))] +
              [(a, 9, b) for a, b in zip(range(144, 256), range(400, 512))] +
              [(a, 7, b) for a, b in zip(range(256, 280), range(0, 24))] +
              [(a, 8, b) for a, b in zip(range(280, 288), range(192, 200))])
    assert oracle.compute_code_values(lengths) == expect


def test_oracle_agrees_with_system_zlib_on_valid_streams(oracle):
    for seed in range(300):
        n = [0, 1, 2, 5, 100, 1000, 5000, 40000, 70000, 200000][seed % 10] if seed % 7 == 0 else (seed * 37) % 20000
        d = corpus.mixed_data(n, seed)
        z = corpus.compress_variant(d, seed)
        r, out = oracle.decompress(z, len(d))
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed


def test_oracle_and_zlib_accept_the_same_corrupt_streams(oracle):
    for seed in range(1500):
        d = corpus.mixed_data((seed * 131) % 3000 + 1, seed)
        z = corpus.corrupt(corpus.compress_variant(d, seed), seed)
        r, out = oracle.decompress(z)
        try:
            dz = zlib.decompressobj()
            oz = dz.decompress(z)
            ok = dz.eof
        except zlib.error:
            ok = False
        assert (r.status == 0) == ok, (seed, r.status, r.message)
        if ok:
            assert out == oz


def test_adler32(oracle):
    for n in (0, 1, 5551, 5552, 5553, 70000):
        b = corpus.random_bytes(n, n)
        assert oracle.adler32(b) == zlib.adler32(b)
    assert oracle.adler32(b"\xff" * 100000, 0xfff0fff0) == zlib.adler32(b"\xff" * 100000, 0xfff0fff0)


def test_chunk_semantics(oracle):
    """Zlib.hs:37-51: whole unread chunks after Done are an error; trailing bytes inside a chunk are not."""
    d = corpus.zipf_text(3000, 1)
    z = zlib.compress(d)
    r, out = oracle.decompress_chunks([z[:100], z[100:200], z[200:]])
    assert r.status == 0 and out == d
    r, out = oracle.decompress_chunks([z + b"junk"])
    assert r.status == 0 and out == d
    r, _ = oracle.decompress_chunks([z, b"junk"])
    assert r.status == oracle.E_DATA_REMAINING
    assert r.message == b"Decompression error: Finished with data remaining."
    r, _ = oracle.decompress_chunks([])
    assert r.status == oracle.E_TRUNCATED


def test_stored_block_chunk_edge_quirk_is_optional(oracle):
    """Monad.hs:280-293: a stored block ending exactly at a chunk end re-requests input and steals a byte.
    The restatement does not do that unless asked."""
    d = corpus.random_bytes(1000, 4)
    z = zlib.compress(d, 0)  # one stored block: 2 header + 5 block header + 1000 data + 4 trailer
    cut = 2 + 5 + 1000
    r, out = oracle.decompress_chunks([z[:cut], z[cut:]])
    assert r.status == 0 and out == d
    r2, out2 = oracle.decompress_chunks([z[:cut], z[cut:]], flags=oracle.F_REF_CHUNK_BUG)
    assert r2.quirks & oracle.QUIRK_STOLEN_BYTE and (r2.status != 0 or out2 != d)


ZLIB_IS_STRICTER = {"dynamic_15bit_codes", "single_distance_code"}


def load_vectors():
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
        return json.load(f)["vectors"]


@pytest.mark.parametrize("v", load_vectors(), ids=lambda v: v["name"])
def test_pinned_generated_vectors(oracle, v):
    z = bytes.fromhex(v["z"])
    r, out = oracle.decompress(z, 1 << 21)
    assert r.status == v["status"] and r.message.decode() == v["message"]
    assert [r.detail0, r.detail1] == v["detail"]
    assert r.out_len == v["out_len"] and hashlib.sha256(out).hexdigest() == v["out_sha256"]
    assert r.in_used == v["in_used"] and r.adler == v["adler"] and r.quirks == v["quirks"]
    if v["status"] == 0:
        # system zlib must agree, except where it is stricter than the reference by design: zlib
        # rejects incomplete code-length codes, which pure-zlib accepts
        if v["name"] in ZLIB_IS_STRICTER:
            with pytest.raises(zlib.error):
                zlib.decompress(z[: r.in_used])
        else:
            assert zlib.decompress(z[: r.in_used]) == out


def test_oracle_gzip_extension_against_system_zlib(oracle):
    """RFC 1952 members (an extension: the reference has no gzip, so this part of the oracle is pinned against
    system zlib, wbits = 31, not against pure-zlib): valid members with every optional header field, then
    single-bit corruptions where the oracle and zlib must agree on accept / reject."""
    import corpus
    assert oracle.crc32(b"123456789") == 0xCBF43926
    for seed in range(120):
        d = corpus.mixed_data((seed * 977) % 40000, seed)
        z = corpus.gzip_member(d, seed)
        assert zlib.decompress(z, 31) == d
        r, o = oracle.gzip_decompress(z, len(d) + 8)
        assert r.status == 0 and o == d and r.adler == zlib.crc32(d) and r.in_used == len(z), (seed, r.status, r.message)
        for c in range(4):
            zz = bytearray(z)
            pos = (seed * 7919 + c * 104729) % len(zz)
            zz[pos] ^= 1 << ((seed + c) % 8)
            r2, o2 = oracle.gzip_decompress(bytes(zz), len(d) + 8)
            try:
                ref = zlib.decompress(bytes(zz), 31)
            except zlib.error:
                ref = None
            if ref is None:
                # (MTIME/XFL/OS/FNAME bytes are not covered by anything unless FHCRC is set: both accept those)
                assert r2.status != 0, (seed, c, pos)
            else:
                assert r2.status == 0 and o2 == ref, (seed, c, pos, r2.status)
