#!/usr/bin/env python3
"""Generates tests/golden/vectors.json: seeded zlib streams for everything the reference's nine
fixtures do not cover: fixed-Huffman blocks, multi-block streams, stored +
compressed mixes, FDICT, CINFO < 7, distance 32768, length 258 / distance codes 28-29, 15-bit codes,
incomplete and empty distance codes, and every error path.

Valid streams come from the system zlib (an independent, specification-equal encoder); the edge
cases zlib never emits are hand-assembled with the little DEFLATE writer below.  Expected outcomes
(status, exact `show` text, length + SHA-256 of the output) are those of oracle/pz_oracle.c at
generation time and are PINNED in the JSON, so a later change of the oracle that alters any of
them fails the test-suite.  Run from the repo root:  python tests/golden/make_vectors.py
"""
import hashlib
import json
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
from oracle import oracle as O  # noqa: E402


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, value, n):  # n bits, LSB first (header fields, extra bits)
        for i in range(n):
            self.bits.append((value >> i) & 1)

    def code(self, code, n):  # a Huffman code: MSB of the code first
        for i in range(n - 1, -1, -1):
            self.bits.append((code >> i) & 1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        self.align()
        out = bytearray()
        for i in range(0, len(self.bits), 8):
            out.append(sum(b << k for k, b in enumerate(self.bits[i:i + 8])))
        return bytes(out)


LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def canonical(lengths):
    """RFC 1951 3.2.2: {symbol: (code, len)} for nonzero lengths."""
    bl = [0] * 17
    for l in lengths:
        bl[l] += 1
    bl[0] = 0
    nxt, code = [0] * 17, 0
    for b in range(1, 17):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[s] = (nxt[l], l)
            nxt[l] += 1
    return out


FIXED_LIT = canonical([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
FIXED_DIST = canonical([5] * 32)


def len_sym(length):
    for i in range(28, -1, -1):
        if LEN_BASE[i] <= length and (i == 28 or length < LEN_BASE[i] + (1 << LEN_EXTRA[i])):
            if i == 28 and length != 258:
                continue
            return 257 + i, length - LEN_BASE[i], LEN_EXTRA[i]
    raise ValueError(length)


def dist_sym(dist):
    for i in range(29, -1, -1):
        if DIST_BASE[i] <= dist:
            return i, dist - DIST_BASE[i], DIST_EXTRA[i]
    raise ValueError(dist)


def emit_tokens(w, tokens, lit, dist):
    """tokens: ints (literal bytes), ('m', len, dist), ('sym', litlen_symbol), ('dsym', len, dist_symbol), 'eob'."""
    for t in tokens:
        if isinstance(t, int):
            w.code(*lit[t])
        elif t == "eob":
            w.code(*lit[256])
        elif t[0] == "sym":
            w.code(*lit[t[1]])
        elif t[0] == "m":
            s, ev, eb = len_sym(t[1])
            w.code(*lit[s])
            w.put(ev, eb)
            d, dv, db = dist_sym(t[2])
            w.code(*dist[d])
            w.put(dv, db)
        elif t[0] == "dsym":
            s, ev, eb = len_sym(t[1])
            w.code(*lit[s])
            w.put(ev, eb)
            w.code(*dist[t[2]])
        elif t[0] == "rawbits":
            w.put(t[1], t[2])
        else:
            raise ValueError(t)


def fixed_block(w, tokens, final=True):
    w.put(1 if final else 0, 1)
    w.put(1, 2)
    emit_tokens(w, tokens, FIXED_LIT, FIXED_DIST)


def stored_block(w, data, final=True, nlen=None):
    w.put(1 if final else 0, 1)
    w.put(0, 2)
    w.align()
    w.put(len(data), 16)
    w.put((~len(data)) & 0xffff if nlen is None else nlen, 16)
    for b in data:
        w.put(b, 8)


ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def dynamic_block(w, lit_lens, dist_lens, tokens, final=True, cl_override=None, clsyms_override=None, hclen=19):
    """A dynamic block whose code lengths are sent literally (no 16/17/18 unless clsyms_override)."""
    w.put(1 if final else 0, 1)
    w.put(2, 2)
    hlit, hdist = len(lit_lens), len(dist_lens)
    w.put(hlit - 257, 5)
    w.put(hdist - 1, 5)
    w.put(hclen - 4, 4)
    clsyms = clsyms_override if clsyms_override is not None else [(l,) for l in lit_lens + dist_lens]
    used = [0] * 19
    for c in clsyms:
        used[c[0]] = 1
    cl_lens = cl_override if cl_override is not None else [5 if u else 0 for u in used]
    if cl_override is None and sum(used) == 1:
        cl_lens[used.index(1)] = 1
    for i in range(hclen):
        w.put(cl_lens[ORDER[i]], 3)
    cl = canonical(cl_lens)
    for c in clsyms:
        w.code(*cl[c[0]])
        if c[0] == 16:
            w.put(c[1], 2)
        elif c[0] == 17:
            w.put(c[1], 3)
        elif c[0] == 18:
            w.put(c[1], 7)
    lit = canonical(lit_lens)
    dist = canonical(dist_lens)
    emit_tokens(w, tokens, lit, dist)


def zwrap(deflate_bytes, data=None, cmf=0x78, flg=None, adler=None, dictid=None):
    if flg is None:
        flg = 0x9c if dictid is None else 0xbb
        flg = (flg & 0xe0) | 0
        rem = ((cmf << 8) | flg) % 31
        if rem:
            flg += 31 - rem
    out = bytes([cmf, flg])
    if dictid is not None:
        out += dictid.to_bytes(4, "big")
    out += deflate_bytes
    if adler is None:
        adler = zlib.adler32(data if data is not None else b"")
    return out + adler.to_bytes(4, "big")


def main():
    rng = random.Random(0xC0DE)
    V = []

    def add(name, z, note):
        r, out = O.decompress(z, 1 << 21)
        V.append({
            "name": name, "note": note, "z": z.hex(),
            "status": int(r.status), "detail": [int(r.detail0), int(r.detail1)], "message": r.message.decode(),
            "out_len": int(r.out_len), "out_sha256": hashlib.sha256(out).hexdigest(), "in_used": int(r.in_used),
            "adler": int(r.adler), "quirks": int(r.quirks),
        })
        return r, out

    # ---- valid streams from the system zlib --------------------------------------------------------
    text = corpus.zipf_text(6000, 11)
    co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    add("fixed_huffman_text", co.compress(text) + co.flush(), "BTYPE=1 via Z_FIXED (reference has no fixed-Huffman fixture)")
    co = zlib.compressobj(6)
    z = b""
    for i in range(0, len(text), 700):
        z += co.compress(text[i:i + 700]) + co.flush(zlib.Z_FULL_FLUSH if i % 1400 else zlib.Z_SYNC_FLUSH)
    add("multi_block_flushes", z + co.flush(), "many dynamic blocks + empty stored blocks from sync/full flushes")
    add("level0_stored_multi", zlib.compress(corpus.random_bytes(70000, 3), 0), "two stored blocks (65535 + rest)")
    for wb in (9, 10, 12, 14):
        co = zlib.compressobj(6, zlib.DEFLATED, wb)
        add(f"cinfo_wbits{wb}", co.compress(text) + co.flush(), f"CINFO={wb - 8}")
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_HUFFMAN_ONLY)
    add("huffman_only", co.compress(text) + co.flush(), "literals only: empty/one-code distance tree in a valid stream")
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    add("rle_runs", co.compress(b"".join(bytes([i]) * (3 + 37 * i % 400) for i in range(64))) + co.flush(), "dist=1 overlaps of many lengths")
    add("empty_output", zlib.compress(b""), "zero-length output")
    add("one_byte", zlib.compress(b"Z"), "one literal")
    add("trailing_garbage_ignored", zlib.compress(text[:500]) + b"\xde\xad\xbe\xef", "Zlib.hs:46-49: trailing bytes inside the chunk are ignored")
    big = corpus.zipf_text(200000, 5)
    add("text_200k_level9", zlib.compress(big, 9), "window slides many times; several blocks")
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY, corpus.zipf_text(2000, 77))
    add("fdict_preset_dictionary", co.compress(corpus.zipf_text(2000, 77) + text[:300]) + co.flush(),
        "FDICT set: the reference skips DICTID and decodes with an empty history (Zlib.hs:68)")

    # ---- hand-assembled valid edge cases -------------------------------------------------------------
    base = corpus.random_bytes(32768, 9)
    w = BitWriter()
    stored_block(w, base, final=False)
    fixed_block(w, [("m", 258, 32768), ("m", 3, 32768), ("m", 200, 24577), ("m", 10, 16385), ("m", 258, 1), "eob"])
    data = bytearray(base)
    for ln, d in ((258, 32768), (3, 32768), (200, 24577), (10, 16385), (258, 1)):
        for _ in range(ln):
            data.append(data[-d])
    add("max_distance_32768_len258", zwrap(w.bytes(), bytes(data)), "distance 32768 (zlib never emits it), length code 285, distance codes 28/29")

    # a complete literal/length code with 15-bit codes: lengths 1,2,...,14,15,15 over chosen symbols
    lit_lens = [0] * 257
    syms = list(range(65, 65 + 15)) + [256]
    for i, s_ in enumerate(syms[:-2]):
        lit_lens[s_] = i + 1
    lit_lens[syms[-2]] = 15   # 'O'
    lit_lens[256] = 15
    lit_lens[syms[13]] = 14
    toks = [65, 66, 67, 78, 79, 79, 78, 77, 65, "eob"]
    w = BitWriter()
    dynamic_block(w, lit_lens, [0], toks, cl_override=None)
    add("dynamic_15bit_codes", zwrap(w.bytes(), bytes([t for t in toks if isinstance(t, int)])),
        "code lengths 1..15: exercises codes longer than the 10-bit primary table; distance tree empty but unused")

    # one distance code of length 1 (incomplete distance code, legal) used by a match
    lit_lens = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 6
    w = BitWriter()
    dynamic_block(w, lit_lens, [1], [97, 98, 99, ("m", 9, 1), "eob"])
    add("single_distance_code", zwrap(w.bytes(), b"abc" + b"c" * 9), "incomplete distance code: one 1-bit code")

    # code-length RLE: 16 right at the start (repeats 0), 17, 18, and a repeat crossing HLIT/HDIST
    lit_lens = [0] * 257
    lit_lens[48] = 1
    lit_lens[256] = 1
    cls = [(18, 48 - 11), (1,), (18, 127), (18, 256 - 49 - 138 + 11 - 11)]
    # positions: 0..47 zeros (18: 48), 48 -> 1, then zeros up to 255, then 256 -> 1 and distance lengths
    cls = [(18, 37), (1,), (18, 127), (18, 69), (1,), (1,), (16, 0)]  # 48 zeros, '0':1, 138+80 zeros... adjusted below
    n_after = 256 - 49
    cls = [(18, 48 - 11), (1,), (18, 138 - 11), (18, n_after - 138 - 11), (1,), (1,), (16, 1)]
    w = BitWriter()
    # HLIT=257, HDIST=5: lengths = 257 lit + 5 dist; the final 16 repeats the dist length 1 four times -> 1,1,1,1,1
    dynamic_block(w, lit_lens, [1, 1, 1, 1, 1], [48, 48, "eob"], clsyms_override=cls)
    add("codelen_rle_16_17_18", zwrap(w.bytes(), b"00"), "code-length symbols 16/18 incl. a run across the HLIT boundary; over-subscribed distance code -> error")

    # ---- error paths (the reference has no negative tests; texts by reading, pinned by the oracle) -------------
    good = zlib.compress(text[:800], 6)
    add("err_truncated_mid", good[:len(good) // 2], "DecompressionError: ran out of data")
    add("err_truncated_trailer", good[:-2], "trailer cut")
    add("err_empty_input", b"", "empty input")
    add("err_one_byte_input", b"\x78", "header cut")
    add("err_header_fcheck", b"\x78\x9d" + good[2:], "FCHECK")
    add("err_header_method", bytes([0x77, 0x9c + ((31 - ((0x77 << 8 | 0x9c) % 31)) % 31)]) + good[2:], "CM != 8")
    cmf = 0x88
    flg = 31 - ((cmf << 8) % 31)
    add("err_header_window", bytes([cmf, flg]) + good[2:], "CINFO = 8")
    bad = bytearray(good)
    bad[-1] ^= 0x55
    add("err_checksum", bytes(bad), "Adler-32 mismatch")
    w = BitWriter()
    stored_block(w, b"\x00" * 1, final=True)
    add("err_checksum_leading_zero_hex", zwrap(w.bytes(), b"\x00", adler=0x0000abcd), "showHex drops leading zeros")
    w = BitWriter()
    w.put(1, 1)
    w.put(3, 2)
    add("err_btype3", zwrap(w.bytes(), b""), "BTYPE=3")
    w = BitWriter()
    stored_block(w, b"hello", nlen=0x1234)
    add("err_len_nlen", zwrap(w.bytes(), b"hello"), "LEN/NLEN mismatch")
    w = BitWriter()
    fixed_block(w, [104, 105, ("m", 3, 3), "eob"])
    add("err_distance_too_far", zwrap(w.bytes(), b"hi"), "distance 3 with 2 bytes produced: the reference throws")
    w = BitWriter()
    fixed_block(w, [104, ("sym", 286), "eob"])
    add("err_litlen_symbol_286", zwrap(w.bytes(), b"h"), "fixed code contains 286/287: the reference throws")
    w = BitWriter()
    fixed_block(w, [104, 105, 106, ("dsym", 3, 30), "eob"])
    add("err_dist_symbol_30", zwrap(w.bytes(), b"hij"), "fixed distance code 30: the reference throws")
    # empty distance tree + a match -> "Tried to advance empty tree!"
    lit_lens = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 6
    w = BitWriter()
    dynamic_block(w, lit_lens, [0], [97, ("sym", 257), ("rawbits", 0, 8), "eob"])
    add("err_empty_distance_tree_used", zwrap(w.bytes(), b"a"), "HuffmanTreeError: advance empty tree")
    # incomplete distance code, unassigned pattern read -> "Advanced to empty tree!"
    w = BitWriter()
    dynamic_block(w, lit_lens, [1], [97, ("sym", 257), ("rawbits", 1, 1), "eob"])
    add("err_unassigned_distance_code", zwrap(w.bytes(), b"a"), "HuffmanTreeError: advanced to empty tree")
    # over-subscribed literal/length code -> insertion errors (three lengths-1 codes)
    ll = [0] * 257
    ll[10] = ll[20] = ll[256] = 1
    w = BitWriter()
    try:
        dynamic_block(w, ll, [1], ["eob"])
    except KeyError:
        pass
    add("err_oversubscribed_litlen", zwrap(w.bytes(), b""), "three 1-bit codes: createHuffmanTree fails")
    ll = [0] * 257
    ll[10] = 1
    ll[20] = 2
    ll[30] = 2
    ll[256] = 2
    w = BitWriter()
    try:
        dynamic_block(w, ll, [1], ["eob"])
    except KeyError:
        pass
    add("err_oversubscribed_litlen_mixed", zwrap(w.bytes(), b""), "1,2,2,2: a different insertion failure")
    ll = [0] * 257
    ll[10] = ll[20] = ll[30] = ll[40] = 2
    ll[256] = 1
    w = BitWriter()
    try:
        dynamic_block(w, ll, [1], ["eob"])
    except KeyError:
        pass
    add("err_oversubscribed_value_hit", zwrap(w.bytes(), b""), "2,2,2,2,1: 'HuffmanValue hit while inserting a value!'")
    # over-subscribed code-length code
    w = BitWriter()
    w.put(1, 1)
    w.put(2, 2)
    w.put(0, 5)
    w.put(0, 5)
    w.put(15, 4)
    for _ in range(19):
        w.put(1, 3)
    add("err_oversubscribed_codelen_code", zwrap(w.bytes(), b""), "nineteen 1-bit code-length codes")
    # seeded bit-flip / truncation / splice corruptions of real streams
    for seed in range(60):
        d = corpus.mixed_data(200 + 53 * seed, seed)
        add(f"err_fuzz_{seed:02d}", corpus.corrupt(corpus.compress_variant(d, seed), seed + 1000), "seeded corruption")

    path = os.path.join(ROOT, "tests", "golden", "vectors.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_vectors.py", "oracle": "oracle/pz_oracle.c", "vectors": V}, f, indent=0)
    st = {}
    for v in V:
        st[v["status"]] = st.get(v["status"], 0) + 1
    print(f"wrote {len(V)} vectors to {path}; statuses {dict(sorted(st.items()))}; {os.path.getsize(path) / 1024:.0f} KiB")
    for v in V:
        if not v["name"].startswith("err_fuzz"):
            print(f'  {v["name"]:34s} status {v["status"]:3d} out {v["out_len"]:7d}  {v["message"][:70]}')


if __name__ == "__main__":
    main()
