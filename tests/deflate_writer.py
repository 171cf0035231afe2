"""TEST INFRASTRUCTURE: a DEFLATE *writer* -- not a compressor.  It emits chosen tokens with chosen code lengths, so that the
decoders meet what zlib's encoder never writes but the reference accepts (reference:
src/Codec/Compression/Zlib/Deflate.hs:124-156 getCodeLengths, HuffmanTree.hs:43-83): 13-15-bit literal/length and distance
codes in long blocks, incomplete codes, a single distance code, code 16 with no predecessor, code-length runs past
HLIT + HDIST, HLIT = 288 / HDIST = 32, matches of 258 bytes and of distance 32768, distance-1 runs, hundreds of tiny dynamic
blocks at odd bit offsets, fixed / stored / dynamic blocks interleaved.

exotic_stream(seed) -> (data, zlib stream, note).  Everything is seeded; nothing here is part of the product."""
import heapq
import random
import zlib

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]

_LEN_SYM = {}
for _i in range(29):
    _hi = 258 if _i == 28 else LEN_BASE[_i] + (1 << LEN_EXTRA[_i]) - 1
    for _l in range(LEN_BASE[_i], _hi + 1):
        _LEN_SYM.setdefault(_l, (257 + _i, _l - LEN_BASE[_i], LEN_EXTRA[_i]))
_LEN_SYM[258] = (285, 0, 0)


def dist_sym(dist):
    lo, hi = 0, 29
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if DIST_BASE[mid] <= dist:
            lo = mid
        else:
            hi = mid - 1
    return lo, dist - DIST_BASE[lo], DIST_EXTRA[lo]


_REV8 = [int(f"{i:08b}"[::-1], 2) for i in range(256)]


def _rev(code, n):  # the n-bit code, bit-reversed (Huffman codes go out MSB first, everything else LSB first)
    return ((_REV8[code & 255] << 8) | _REV8[(code >> 8) & 255]) >> (16 - n)


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        self.acc |= value << self.n
        self.n += nbits
        if self.n >= 64:
            k = self.n // 8
            self.out += (self.acc & ((1 << (8 * k)) - 1)).to_bytes(k, "little")
            self.acc >>= 8 * k
            self.n -= 8 * k

    def code(self, c):  # c = (code, length): already bit-reversed by canonical()
        self.put(c[0], c[1])

    def align(self):
        self.put(0, (-self.n) % 8)

    def bitpos(self):
        return 8 * len(self.out) + self.n

    def bytes(self):
        self.align()
        k = self.n // 8
        return bytes(self.out + self.acc.to_bytes(k, "little"))


def canonical(lengths):
    """RFC 1951 3.2.2 (= Deflate.hs:255-292 computeCodeValues): {symbol: (bit-reversed code, length)} for nonzero lengths."""
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
            out[s] = (_rev(nxt[l], l), l)
            nxt[l] += 1
    return out


FIXED_LIT = canonical([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
FIXED_DIST = canonical([5] * 32)


def kraft(lengths):
    return sum(1 << (15 - l) for l in lengths if l)  # in units of 2^-15


def huffman_lengths(freq, limit):
    """Code lengths of a Huffman code for the symbols with freq > 0, no longer than `limit` (frequencies are flattened until it fits)."""
    f = list(freq)
    while True:
        heap = [(w, i, (i,)) for i, w in enumerate(f) if w > 0]
        if len(heap) == 1:
            l = [0] * len(f)
            l[heap[0][1]] = 1
            return l
        heapq.heapify(heap)
        depth = [0] * len(f)
        tie = len(f)
        while len(heap) > 1:
            a = heapq.heappop(heap)
            b = heapq.heappop(heap)
            for s in a[2] + b[2]:
                depth[s] += 1
            heapq.heappush(heap, (a[0] + b[0], tie, a[2] + b[2]))
            tie += 1
        if max(depth) <= limit:
            return depth
        f = [(w + 1) // 2 if w > 0 else 0 for w in f]


def deep_lengths(freq, rng, limit=15, tail=None, complete=False):
    """Lengths that make LONG codes common: the most frequent symbols get 2, 3, 4, ... bits, every other used symbol `tail`
    (13-15) bits -- an incomplete code, which the reference accepts (HuffmanTree.hs:43-83); complete=True hands the unused
    code space to symbols that never occur."""
    tail = tail or rng.choice([13, 14, 15])
    used = sorted((i for i, w in enumerate(freq) if w > 0), key=lambda i: -freq[i])
    l = [0] * len(freq)
    nxt = 2
    for rank, s in enumerate(used):
        if nxt < tail - 1 and rank < tail - 3:
            l[s] = nxt
            nxt += 1
        else:
            l[s] = tail
    while kraft(l) > 32768:  # (too many symbols for this profile: push the short ones down)
        s = min((i for i in used if l[i] < limit), key=lambda i: l[i])
        l[s] += 1
    if complete:
        spare = [i for i in range(len(freq)) if l[i] == 0]
        rng.shuffle(spare)
        for s in spare:
            room = 32768 - kraft(l)
            if room <= 0:
                break
            b = max(1, 15 - (room.bit_length() - 1))
            l[s] = b if (1 << (15 - b)) <= room else 15
    return l


def encode_lengths(seq, rng, style):
    """The code-length alphabet's symbols for the length sequence `seq` (HLIT + HDIST lengths as ONE sequence, Deflate.hs:124-156):
    [(sym,), (16, extra), (17, extra), (18, extra)].  style: 'plain' (no repeats), 'rle' (greedy), 'mixed' (seeded choices),
    'overrun' (the last zero run reaches past the end of the sequence: the reference accepts it, Deflate.hs:132)."""
    out, i, n = [], 0, len(seq)
    prev = None
    while i < n:
        v = seq[i]
        run = 1
        while i + run < n and seq[i + run] == v:
            run += 1
        use_rle = style != "plain" and (style != "mixed" or rng.random() < 0.7)
        if v == 0 and use_rle and run >= 3:
            take = min(run, 138)
            if style == "mixed":
                take = rng.randint(3, take)
            if style == "overrun" and i + take == n:
                over = rng.randint(1, 20)
                if take + over <= 10 or 11 <= take + over <= 138:
                    take += over
            if take <= 10:
                out.append((17, take - 3))
            else:
                out.append((18, take - 11))
            i += min(take, n - i)
            prev = 0
            continue
        if v != 0 and use_rle and prev == v and run >= 3:
            take = min(run, 6)
            out.append((16, take - 3))
            i += take
            continue
        if v == 0 and prev is None and style in ("mixed", "overrun") and run >= 3 and rng.random() < 0.5:
            take = min(run, 6)  # code 16 with no predecessor repeats 0 (Deflate.hs:91,136-139)
            out.append((16, take - 3))
            i += take
            prev = 0
            continue
        out.append((v,))
        prev = v
        i += 1
    return out


class Block:
    """One DEFLATE block in the making: tokens are ints (literal bytes) and (length, distance) pairs."""

    def __init__(self, kind):
        self.kind = kind  # 'dynamic' | 'fixed' | 'stored'
        self.tokens = []
        self.raw = b""


def write_block(w, blk, final, rng, opts):
    w.put(1 if final else 0, 1)
    if blk.kind == "stored":
        w.put(0, 2)
        w.align()
        w.put(len(blk.raw), 16)
        w.put((~len(blk.raw)) & 0xffff, 16)
        for b in blk.raw:
            w.put(b, 8)
        return
    if blk.kind == "fixed":
        w.put(1, 2)
        lit, dist = FIXED_LIT, FIXED_DIST
    else:
        w.put(2, 2)
        lf, df = [0] * 286, [0] * 30
        lf[256] = 1
        for t in blk.tokens:
            if isinstance(t, int):
                lf[t] += 1
            else:
                lf[_LEN_SYM[t[0]][0]] += 1
                df[dist_sym(t[1])[0]] += 1
        style = opts.get("codes", "huffman")
        if style == "huffman":
            ll = huffman_lengths(lf, 15)
            dl = huffman_lengths(df, 15) if any(df) else [0] * 30
        else:
            ll = deep_lengths(lf, rng, complete=(style == "deep_complete"))
            dl = deep_lengths(df, rng, complete=(style == "deep_complete")) if any(df) else [0] * 30
        if opts.get("single_dist") and any(df):  # ONE distance code of one bit: incomplete, accepted (HuffmanTree.hs)
            assert sum(1 for x in df if x) == 1
            dl = [1 if x else 0 for x in df]
        if not any(dl):
            dl = [0] * 30
            if opts.get("empty_dist_ok") is not True:
                dl[0] = 1  # (zlib's habit: one unused one-bit code; the reference accepts an all-zero tree too -- vectors.json pins that)
        hl = opts.get("hlit")
        if hl is None:
            hl = max(257, max(i for i, x in enumerate(ll) if x) + 1)
            if rng.random() < 0.3:
                hl = rng.randint(hl, 286)
        ll = (ll + [0, 0])[:max(hl, 257)] if hl > len(ll) else ll[:hl]
        hd = opts.get("hdist")
        if hd is None:
            hd = max(1, max((i for i, x in enumerate(dl) if x), default=0) + 1)
            if rng.random() < 0.3:
                hd = rng.randint(hd, 30)
        dl = (dl + [0, 0])[:hd]
        seq = ll + dl
        cls = encode_lengths(seq, rng, opts.get("rle", "rle"))
        cf = [0] * 19
        for c in cls:
            cf[c[0]] += 1
        cl_lens = huffman_lengths(cf, 7) if opts.get("cl_huffman", True) else [5 if x else 0 for x in cf]
        if sum(1 for x in cl_lens if x) == 1:
            cl_lens[cl_lens.index(max(cl_lens))] = 1
        hclen = 19
        if opts.get("trim_hclen", True):
            while hclen > 4 and cl_lens[ORDER[hclen - 1]] == 0:
                hclen -= 1
        w.put(len(ll) - 257, 5)
        w.put(len(dl) - 1, 5)
        w.put(hclen - 4, 4)
        for i in range(hclen):
            w.put(cl_lens[ORDER[i]], 3)
        cl = canonical(cl_lens)
        for c in cls:
            w.code(cl[c[0]])
            if c[0] == 16:
                w.put(c[1], 2)
            elif c[0] == 17:
                w.put(c[1], 3)
            elif c[0] == 18:
                w.put(c[1], 7)
        lit, dist = canonical(ll), canonical(dl)
    for t in blk.tokens:
        if isinstance(t, int):
            w.code(lit[t])
        else:
            s, ev, eb = _LEN_SYM[t[0]]
            w.code(lit[s])
            w.put(ev, eb)
            d, dv, db = dist_sym(t[1])
            w.code(dist[d])
            w.put(dv, db)
    w.code(lit[256])


def gen_tokens(rng, out, nbytes, prof):
    """Appends ~nbytes of output to the bytearray `out` and returns the tokens that produce them.  prof: p_match, alphabet
    (literal byte pool), lens (length pool), dists ('near' | 'far' | 'any' | 'one' | 'max' | a fixed distance)."""
    tokens = []
    target = len(out) + nbytes
    alphabet, lens, dmode, pm = prof["alphabet"], prof["lens"], prof["dists"], prof["p_match"]
    while len(out) < target:
        if out and rng.random() < pm:
            ln = rng.choice(lens)
            have = min(len(out), 32768)
            if dmode == "one":
                d = 1
            elif dmode == "max":
                d = have if rng.random() < 0.5 else rng.randint(max(1, have - 300), have)
            elif dmode == "near":
                d = rng.randint(1, min(have, 300))
            elif dmode == "far":
                d = rng.randint(min(have, 2000), have) if have > 2000 else rng.randint(1, have)
            elif isinstance(dmode, int):
                if have < dmode:  # (the one distance this block uses is not there yet: a literal instead)
                    b = rng.choice(alphabet)
                    tokens.append(b)
                    out.append(b)
                    continue
                d = dmode
            else:
                d = rng.randint(1, have) if rng.random() < 0.5 else rng.randint(1, min(have, 1 << rng.randint(0, 15)))
            tokens.append((ln, d))
            start = len(out) - d
            if d >= ln:
                out += out[start:start + ln]
            else:
                for k in range(ln):
                    out.append(out[start + k])
        else:
            b = rng.choice(alphabet)
            tokens.append(b)
            out.append(b)
    return tokens


def _alphabet(rng, kind):
    if kind == "text":
        return list(b"etaoin shrdlucmfwypvbgkqjxz ETAOIN.,;\n") * 4 + list(range(32, 127))
    if kind == "wide":  # many distinct symbols, skewed: long codes in real use
        pool = []
        for i in range(256):
            pool += [i] * max(1, 4000 // (1 + i * i // 8))
        return pool
    if kind == "four":
        return list(b"acgt")
    return list(range(256))


def exotic_stream(seed):
    """(data, zlib stream, note): a valid stream no zlib encoder would write.  The mode cycles with the seed."""
    rng = random.Random(0xE0 + seed * 7919)
    mode = seed % 8
    out = bytearray()
    blocks = []
    note = ""

    def dyn(nbytes, prof, **opts):
        b = Block("dynamic")
        b.tokens = gen_tokens(rng, out, nbytes, prof)
        b.opts = opts
        blocks.append(b)

    def fixed(nbytes, prof):
        b = Block("fixed")
        b.tokens = gen_tokens(rng, out, nbytes, prof)
        b.opts = {}
        blocks.append(b)

    def stored(nbytes):
        b = Block("stored")
        b.raw = bytes(rng.getrandbits(8) for _ in range(nbytes))
        out.extend(b.raw)
        b.opts = {}
        blocks.append(b)

    all_lens = list(range(3, 259)) + [3, 4, 5, 6, 7, 8] * 120  # every length occurs, the short ones most often
    short_lens = [3, 3, 3, 4, 4, 5, 6, 7, 8, 10, 12, 17, 25, 40]
    if mode == 0:
        note = "one long block, 13-15-bit literal/length and distance codes in constant use (incomplete codes)"
        dyn(rng.choice([36000, 52000, 80000]), dict(alphabet=_alphabet(rng, "wide"), lens=all_lens, dists="any", p_match=0.2), codes="deep", rle="mixed")
    elif mode == 1:
        note = "long blocks with complete deep codes (the spare code space on symbols that never occur), HLIT 286 / HDIST 30"
        for _ in range(rng.randint(1, 3)):
            dyn(rng.choice([20000, 34000]), dict(alphabet=_alphabet(rng, "wide"), lens=short_lens, dists="far", p_match=0.35), codes="deep_complete", rle="rle",
                hlit=286, hdist=30)
    elif mode == 2:
        note = "hundreds of tiny dynamic blocks at odd bit offsets (some fixed, some stored), then one long block"
        for _ in range(rng.randint(150, 400)):
            r = rng.random()
            if r < 0.75:
                dyn(rng.randint(1, 120), dict(alphabet=_alphabet(rng, "text"), lens=short_lens, dists="any", p_match=0.3), codes=rng.choice(["huffman", "deep"]),
                    rle=rng.choice(["plain", "rle", "mixed", "overrun"]), cl_huffman=rng.random() < 0.5, trim_hclen=rng.random() < 0.5)
            elif r < 0.9:
                fixed(rng.randint(1, 80), dict(alphabet=_alphabet(rng, "text"), lens=short_lens, dists="near", p_match=0.3))
            else:
                stored(rng.randint(0, 40))
        dyn(34000, dict(alphabet=_alphabet(rng, "text"), lens=short_lens, dists="any", p_match=0.4), codes="huffman", rle="overrun")
    elif mode == 3:
        note = "matches of 258 bytes, distance 1 runs, distances at and next to 32768, in long blocks"
        dyn(34000, dict(alphabet=_alphabet(rng, "text"), lens=[258, 258, 257, 3, 4, 130], dists="one", p_match=0.03), codes="huffman", rle="rle")
        dyn(60000, dict(alphabet=_alphabet(rng, "wide"), lens=[258, 3, 3, 4, 5, 5, 6, 64, 65, 33, 32, 31], dists="max", p_match=0.12), codes="deep", rle="mixed")
    elif mode == 4:
        note = "a single distance code of one bit (incomplete), literal-heavy long block, code 16 first, runs past HLIT + HDIST"
        dyn(rng.choice([40000, 60000]), dict(alphabet=_alphabet(rng, "wide"), lens=short_lens, dists=rng.choice([1, 4, 24, 300, 5000]), p_match=0.08), codes="deep",
            single_dist=True, rle="overrun")
    elif mode == 5:
        note = "fixed, stored and dynamic blocks interleaved, each long enough for the strips; HLIT = 288, HDIST = 32"
        for k in range(rng.randint(3, 6)):
            r = k % 3
            if r == 0:
                fixed(rng.choice([9000, 20000]), dict(alphabet=_alphabet(rng, "text"), lens=short_lens, dists="any", p_match=0.35))
            elif r == 1:
                stored(rng.randint(1, 3000))
            else:
                dyn(rng.choice([12000, 30000]), dict(alphabet=_alphabet(rng, "wide"), lens=all_lens, dists="any", p_match=0.3), codes="deep", hlit=288, hdist=32,
                    rle="mixed", cl_huffman=False)
    elif mode == 6:
        note = "two-bit codes (four symbols) with long matches: the shortest strips, and a tail of tiny blocks"
        dyn(50000, dict(alphabet=_alphabet(rng, "four"), lens=[3, 4, 5, 6, 9, 258], dists="any", p_match=0.05), codes="huffman", rle="plain")
        for _ in range(60):
            dyn(rng.randint(1, 30), dict(alphabet=_alphabet(rng, "four"), lens=[3, 4], dists="near", p_match=0.2), codes="deep", rle="overrun")
    else:
        note = "every byte value, matches rare, deep incomplete codes: the long-code paths all the time; then text with near matches"
        dyn(30000, dict(alphabet=_alphabet(rng, "any"), lens=short_lens, dists="any", p_match=0.03), codes="deep", rle="mixed")
        dyn(30000, dict(alphabet=_alphabet(rng, "text"), lens=short_lens, dists="near", p_match=0.5), codes="huffman", rle="rle")
    w = BitWriter()
    for i, b in enumerate(blocks):
        write_block(w, b, i == len(blocks) - 1, rng, b.opts)
    data = bytes(out)
    z = bytes([0x78, 0x9c]) + w.bytes() + zlib.adler32(data).to_bytes(4, "big")
    return data, z, note


def pool_stream(seed):
    """(data, zlib stream): dynamic blocks long enough for the kernels' spans whose literal/length code has 150-256 literals of 8 to
    12 bits in constant use -- codes that need a little less, a little more or much more than the 252 second-level entries the kernels'
    pool holds (round 6: the depth cap's remainder goes to the first prefixes it cut short; long codes are resolved inside the spans)."""
    rng = random.Random(0x9001 + seed * 104729)
    out = bytearray()
    w = BitWriter()
    nblocks = 1 + seed % 3
    for bi in range(nblocks):
        k = rng.choice([150, 180, 200, 220, 240, 256])
        syms = rng.sample(range(256), k)
        # a few frequent symbols (short codes), a broad middle (8-9 bits), a tail two to sixteen times rarer (10-12 bits)
        hot = rng.randint(0, 6)
        tail = rng.randint(0, k // 2)
        pool = []
        for i, sy in enumerate(syms):
            wgt = 400 if i < hot else (rng.choice([1, 2, 4]) if i >= k - tail else 16)
            pool += [sy] * wgt
        b = Block("dynamic")
        b.tokens = gen_tokens(rng, out, rng.choice([9000, 20000, 45000]),
                              dict(alphabet=pool, lens=[3, 3, 4, 4, 5, 6, 8, 12, 30, 100], dists=rng.choice(["near", "any", "far", 16]),
                                   p_match=rng.choice([0.0, 0.1, 0.3, 0.5])))
        b.opts = {}
        write_block(w, b, bi == nblocks - 1, rng, dict(codes="huffman", rle="rle"))
    d = bytes(out)
    import zlib
    return d, bytes([0x78, 0x9c]) + w.bytes() + zlib.adler32(d).to_bytes(4, "big")
