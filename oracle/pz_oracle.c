/*
 * pz_oracle.c -- plain-C restatement of pure-zlib's decompress path.
 *
 * TEST INFRASTRUCTURE ONLY (see pz_oracle.h).  Parity: PINNED against the
 * reference's nine .z/.gold fixtures and its two computeCodeValues KATs.
 *
 * The structure deliberately mirrors the reference module by module so each
 * piece can be checked against the Haskell it restates:
 *   bit/byte reader ........ src/Codec/Compression/Zlib/Monad.hs:185-307
 *   Huffman trie ........... src/Codec/Compression/Zlib/HuffmanTree.hs:11-83
 *   canonical codes ........ src/Codec/Compression/Zlib/Deflate.hs:255-292
 *   block decode ........... src/Codec/Compression/Zlib/Deflate.hs:39-156
 *   length/distance tables . src/Codec/Compression/Zlib/Deflate.hs:160-237
 *   output window .......... src/Codec/Compression/Zlib/OutputWindow.hs:29-114
 *   Adler-32 ............... src/Codec/Compression/Zlib/Adler32.hs:17-57
 *   container + driver ..... src/Codec/Compression/Zlib.hs:29-69
 *
 * Like the reference it walks the Huffman trie one bit at a time; it is a
 * faithful scalar port, not a fast inflater.
 */
#include "pz_oracle.h"

#include <setjmp.h>
#include <stdio.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* HuffmanTree.hs:11-15  data HuffmanTree a = HuffmanNode l r | HuffmanValue a | HuffmanEmpty */

enum { K_EMPTY = 0, K_VALUE = 1, K_NODE = 2 };

typedef struct {
    uint8_t kind;
    int16_t l, r; /* children (node indices) for K_NODE */
    int16_t val;  /* symbol for K_VALUE */
} hnode;

#define POOL 8192 /* >= 458 symbols * 15 levels + 1 */

typedef struct {
    hnode n[POOL];
    int used; /* n[0] is the shared HuffmanEmpty */
    int root;
} htree;

static void tree_init(htree *t)
{
    t->n[0].kind = K_EMPTY;
    t->used = 1;
    t->root = 0; /* emptyHuffmanTree, HuffmanTree.hs:22-23 */
}

static int tree_new(htree *t, int kind, int l, int r, int val)
{
    int i = t->used++;
    t->n[i].kind = (uint8_t)kind;
    t->n[i].l = (int16_t)l;
    t->n[i].r = (int16_t)r;
    t->n[i].val = (int16_t)val;
    return i;
}

/* HuffmanTree.hs:36-71 addHuffmanNode.  Returns the (possibly new) node index, or -1 with
 * *err set to PZO_INS_*.  `code` is used through testBit only, i.e. its low `len` bits. */
static int add_huffman_node(htree *t, int node, int val, int len, int code, int *err)
{
    hnode *nd = &t->n[node];
    switch (nd->kind) {
    case K_EMPTY:
        if (len == 0) /* :44-46 */
            return tree_new(t, K_VALUE, 0, 0, val);
        { /* :47-52 */
            int child = add_huffman_node(t, 0, val, len - 1, code, err);
            if (child < 0) return -1;
            if ((code >> (len - 1)) & 1)
                return tree_new(t, K_NODE, 0, child, 0);
            return tree_new(t, K_NODE, child, 0, 0);
        }
    case K_VALUE:
        *err = (len == 0) ? PZO_INS_TWO_VALUES /* :54-56 */ : PZO_INS_VALUE_HIT /* :57-58 */;
        return -1;
    default: /* K_NODE */
        if (len == 0) { /* :60-62 */
            *err = PZO_INS_LEAF_IS_NODE;
            return -1;
        }
        if ((code >> (len - 1)) & 1) { /* :63-66 */
            int r = add_huffman_node(t, nd->r, val, len - 1, code, err);
            if (r < 0) return -1;
            t->n[node].r = (int16_t)r;
        } else { /* :67-70 */
            int l = add_huffman_node(t, nd->l, val, len - 1, code, err);
            if (l < 0) return -1;
            t->n[node].l = (int16_t)l;
        }
        return node;
    }
}

/* ------------------------------------------------------------------------- */
/* Deflate.hs:261-288 computeCodeValues (RFC 1951 3.2.2 steps 1-3).
 * in: n (symbol,length) pairs, symbols distinct.  out: ascending by symbol, zero lengths dropped. */

#define MAXSYM 512

int pzo_compute_code_values(const int *syms, const int *lens, int n,
                            int *out_sym, int *out_len, int *out_code)
{
    int len_of[MAXSYM];
    int bl_count[64];
    int next_code[64];
    int i, m = 0, max_bits = 0;

    for (i = 0; i < MAXSYM; i++) len_of[i] = 0;
    memset(bl_count, 0, sizeof bl_count);
    memset(next_code, 0, sizeof next_code);
    /* valsNo0s (:264) and lenTree (:268) */
    for (i = 0; i < n; i++) {
        if (lens[i] != 0 && syms[i] >= 0 && syms[i] < MAXSYM) {
            len_of[syms[i]] = lens[i];
            bl_count[lens[i] & 63]++; /* blCount (:266) */
            if (lens[i] > max_bits) max_bits = lens[i]; /* maxBits (:270) */
        }
    }
    /* step2 (:273-278): nextcode[bits] = (nextcode[bits-1] + blCount[bits-1]) << 1, blCount[0] absent */
    {
        int code = 0, bits;
        bl_count[0] = 0;
        for (bits = 1; bits <= max_bits && bits < 64; bits++) {
            code = (code + bl_count[bits - 1]) << 1;
            next_code[bits] = code;
        }
    }
    /* step3 (:280-288) in ascending symbol order (valsSort, :265) */
    for (i = 0; i < MAXSYM; i++) {
        int len = len_of[i];
        if (len == 0) continue;
        out_sym[m] = i;
        out_len[m] = len;
        out_code[m] = next_code[len & 63]++;
        m++;
    }
    return m;
}

/* ------------------------------------------------------------------------- */
/* Monad.hs:68-74 DecompressionState, flattened; chunks are the lazy ByteString's strict chunks. */

typedef struct {
    const uint8_t *base;
    const uint64_t *coff;
    uint32_t nchunks;
    uint32_t chunk_next;     /* first chunk not yet handed to the decoder (Zlib.hs:40-42) */
    const uint8_t *p, *pend; /* dcsInput */
    int bitno;               /* dcsNextBitNo */
    uint8_t cur;             /* dcsCurByte */
    uint32_t a, b;           /* dcsAdler32 */
    uint32_t crc;            /* PZO_F_GZIP only: CRC-32 register over the bytes produced (extension) */
    uint8_t *out;
    uint64_t cap;
    uint64_t total;          /* bytes produced so far */
    uint64_t ow_next;        /* owNext of the reference's 128 KiB window (emulated cursor only) */
    uint8_t win[65536];      /* history for back-references, independent of the caller's capacity */
    uint32_t flags;
    pzo_result *res;
    jmp_buf jb;
    /* event trace of the incremental protocol (pzo_trace): NeedMore / Chunk / Done / DecompError, Monad.hs:163-167 */
    int32_t *ev_type;
    uint32_t *ev_val;
    uint32_t ev_cap, ev_n;
    /* extension (PZO dictionary): preset dictionary = history in front of the output */
    const uint8_t *dict;
    uint64_t dict_len;
    htree fixed_lit, fixed_dist, code_tree, lit_tree, dist_tree;
} dstate;

/* Monad.hs:152-154 raise + Monad.hs:95-102 show */
static void raise_err(dstate *s, int status, uint32_t d0, uint32_t d1, const char *msg)
{
    s->res->status = status;
    s->res->detail0 = d0;
    s->res->detail1 = d1;
    snprintf(s->res->message, sizeof s->res->message, "%s", msg);
    longjmp(s->jb, 1);
}

static void trace_event(dstate *s, int32_t type, uint32_t val)
{
    if (s->ev_type && s->ev_n < s->ev_cap) {
        s->ev_type[s->ev_n] = type;
        s->ev_val[s->ev_n] = val;
    }
    if (s->ev_type) s->ev_n++;
}

/* Monad.hs:185-197 getNextChunk/loadChunk driven by Zlib.hs:38-42 */
static void get_next_chunk(dstate *s)
{
    for (;;) {
        trace_event(s, PZO_EV_NEED_MORE, 0); /* Monad.hs:186 / :194: the decoder hands NeedMore back */
        if (s->chunk_next >= s->nchunks)
            raise_err(s, PZO_E_TRUNCATED, 0, 0,
                      "Decompression error: Ran out of data mid-decompression 2.");
        {
            const uint8_t *cb = s->base + s->coff[s->chunk_next];
            const uint8_t *ce = s->base + s->coff[s->chunk_next + 1];
            s->chunk_next++;
            if (cb == ce) continue; /* S.uncons = Nothing -> NeedMore again (:194) */
            s->bitno = 0;
            s->cur = *cb;
            s->p = cb + 1;
            s->pend = ce;
            return;
        }
    }
}

/* Monad.hs:203-230 nextBits / nextBits': LSB-first within each byte, fields little-endian */
static uint32_t next_bits(dstate *s, int x)
{
    uint32_t acc = 0;
    int shift = 0;
    while (x != 0) {
        if (s->bitno == 8) { /* :213-220 */
            if (s->p == s->pend) {
                get_next_chunk(s);
            } else {
                s->cur = *s->p++;
                s->bitno = 0;
            }
            continue;
        }
        { /* :221-230 */
            int my = x < (8 - s->bitno) ? x : (8 - s->bitno);
            uint32_t basev = (uint32_t)s->cur >> s->bitno;
            uint32_t mask = ~(0xFFu << my) & 0xFFu;
            acc |= (basev & mask) << shift;
            s->bitno += my;
            x -= my;
            shift += my;
        }
    }
    return acc;
}

/* Monad.hs:232-249 nextByte */
static uint32_t next_byte(dstate *s)
{
    for (;;) {
        if (s->bitno == 0) { /* :236-238 */
            s->bitno = 8;
            return s->cur;
        }
        if (s->bitno != 8) /* :239 */
            return next_bits(s, 8);
        if (s->p == s->pend) { /* :241 */
            get_next_chunk(s);
            continue;
        }
        s->cur = *s->p++; /* :242-249 */
        s->bitno = 8;
        return s->cur;
    }
}

/* Monad.hs:251-255 nextWord16 (little-endian) */
static uint32_t next_word16(dstate *s)
{
    uint32_t low = next_byte(s);
    uint32_t high = next_byte(s);
    return (high << 8) | low;
}

/* Monad.hs:257-263 nextWord32 (BIG-endian) */
static uint32_t next_word32(dstate *s)
{
    uint32_t a = next_byte(s);
    uint32_t b = next_byte(s);
    uint32_t c = next_byte(s);
    uint32_t d = next_byte(s);
    return (a << 24) | (b << 16) | (c << 8) | d;
}

/* Monad.hs:304-307 advanceToByte */
static void advance_to_byte(dstate *s) { s->bitno = 8; }

/* ------------------------------------------------------------------------- */
/* Adler32.hs */

#define ADLER_MOD 65521u

/* Adler32.hs:22-27 advanceAdler */
static void adler_byte(uint32_t *a, uint32_t *b, uint8_t v)
{
    *a = (*a + v) % ADLER_MOD;
    *b = (*b + *a) % ADLER_MOD;
}

/* Adler32.hs:37-51 advanceAdlerBlock: spans of at most 5551 bytes with one modulo at the end */
static void adler_block(uint32_t *pa, uint32_t *pb, const uint8_t *buf, uint64_t len)
{
    uint64_t a = *pa, b = *pb;
    while (len > 0) {
        uint64_t span = len < 5552 ? len : 5551; /* :45-51 */
        uint64_t i;
        for (i = 0; i < span; i++) { /* advanceNoMod :29-34 */
            a += buf[i];
            b += a;
        }
        a %= ADLER_MOD; /* advanceAdlerLimited :40-42 */
        b %= ADLER_MOD;
        buf += span;
        len -= span;
    }
    *pa = (uint32_t)a;
    *pb = (uint32_t)b;
}

uint32_t pzo_adler32(uint32_t adler, const uint8_t *buf, uint64_t len)
{
    uint32_t a = adler & 0xffffu, b = adler >> 16;
    adler_block(&a, &b, buf, len);
    return (b << 16) | a; /* finalizeAdler :53-57 */
}

/* ------------------------------------------------------------------------- */
/* Output side: Monad.hs:309-347 over OutputWindow.hs */

/* The reference's window is a flat 128 KiB vector whose writes are bounds-checked
 * (OutputWindow.hs:29-30,64-68); it throws instead of returning Left when the cursor
 * would run past it.  The restatement only tracks the cursor to report that. */
static void ow_advance(dstate *s, uint64_t n)
{
    if (s->ow_next + n > 128u * 1024u)
        s->res->quirks |= PZO_QUIRK_REF_WINDOW_OVERFLOW;
    s->ow_next += n;
}

/* RFC 1952 section 8, bit by bit (extension: gzip is not in the reference) */
static uint32_t crc32_byte(uint32_t reg, uint8_t v)
{
    int k;
    reg ^= v;
    for (k = 0; k < 8; k++) reg = (reg >> 1) ^ (0xedb88320u & (0u - (reg & 1u)));
    return reg;
}

uint32_t pzo_crc32(uint32_t crc, const uint8_t *buf, uint64_t len)
{
    uint32_t reg = ~crc;
    uint64_t i;
    for (i = 0; i < len; i++) reg = crc32_byte(reg, buf[i]);
    return ~reg;
}

static void put_byte(dstate *s, uint8_t v)
{
    if (s->flags & PZO_F_GZIP) s->crc = crc32_byte(s->crc, v);
    if (s->total < s->cap) s->out[s->total] = v;
    s->win[s->total & 65535u] = v;
    s->total++;
}

/* Monad.hs:309-315 emitByte -> OutputWindow.hs:64-68 addByte + Adler32.hs:22-27 */
static void emit_byte(dstate *s, uint8_t v)
{
    ow_advance(s, 1);
    put_byte(s, v);
    adler_byte(&s->a, &s->b, v);
}

/* Monad.hs:335-347 moveWindow -> OutputWindow.hs:45-54 emitExcess: ONE 32 KiB piece per call */
static void move_window(dstate *s)
{
    if (s->ow_next >= 2u * 32768u) {
        s->ow_next -= 32768u;
        trace_event(s, PZO_EV_CHUNK, 32768u); /* Monad.hs:346 publish builtChunk */
    }
}

/* Monad.hs:324-333 emitPastChunk -> OutputWindow.hs:82-101 addOldChunk/copyChunked:
 * sequential memcpy pieces of <= dist bytes == byte-serial overlapping LZ77 copy. */
static void emit_past_chunk(dstate *s, uint32_t dist, uint32_t len)
{
    uint32_t a, b, i;
    if ((uint64_t)dist > s->total + s->dict_len) { /* MV.slice (next - dist) ... with next - dist < 0 */
        char m[160];
        snprintf(m, sizeof m,
                 "(reference throws) back-reference distance %u exceeds the %llu bytes produced",
                 dist, (unsigned long long)s->total);
        raise_err(s, PZO_E_BAD_DISTANCE, dist, (uint32_t)s->total, m);
    }
    ow_advance(s, len);
    a = s->a;
    b = s->b;
    for (i = 0; i < len; i++) {
        uint8_t v = (uint64_t)dist > s->total ? s->dict[s->dict_len - (dist - s->total)] /* extension: into the preset dictionary */
                                              : s->win[(s->total - dist) & 65535u];
        put_byte(s, v);
        a += v; /* Adler over the copied bytes, Monad.hs:331; len <= 258 < 5552 */
        b += a;
    }
    s->a = a % ADLER_MOD;
    s->b = b % ADLER_MOD;
}

/* Monad.hs:265-293 nextBlock + Monad.hs:317-322 emitBlock (stored data) */
static void emit_stored(dstate *s, uint32_t len)
{
    /* nextBlock is entered with dcsNextBitNo == 8 (nextWord16 leaves it there, :236-249) */
    ow_advance(s, len);
    for (;;) {
        uint64_t have = (uint64_t)(s->pend - s->p);
        if (len < have) { /* getBlock :275-279 */
            uint32_t i;
            adler_block(&s->a, &s->b, s->p, len);
            for (i = 0; i < len; i++) put_byte(s, s->p[i]);
            s->p += len;
            s->bitno = 8;
            return;
        }
        if (have == 0) { /* :280-285 */
            if (len == 0 && !(s->flags & PZO_F_REF_CHUNK_BUG)) {
                /* The reference re-requests input here even though nothing is needed and
                 * appends one stolen byte.  Not replicated unless asked. */
                s->bitno = 8;
                return;
            }
            get_next_chunk(s);
            if (len == 0) { /* literal reference behaviour: steals byte1, getBlock (-1) */
                s->res->quirks |= PZO_QUIRK_STOLEN_BYTE;
                adler_byte(&s->a, &s->b, s->cur);
                put_byte(s, s->cur);
                s->bitno = 8;
                return;
            }
            adler_byte(&s->a, &s->b, s->cur);
            put_byte(s, s->cur);
            len -= 1;
            continue; /* getBlock (len-1) (dcsInput dcs); bitno is reset on the final split */
        }
        /* otherwise :286-288: take the whole remaining chunk, continue with S.empty */
        {
            uint64_t i;
            adler_block(&s->a, &s->b, s->p, have);
            for (i = 0; i < have; i++) put_byte(s, s->p[i]);
            s->p += have;
            len -= (uint32_t)have;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Deflate.hs:255-259 computeHuffmanTree = createHuffmanTree . computeCodeValues
 * HuffmanTree.hs:25-34 createHuffmanTree = foldr: the LAST triple is inserted first. */
static void compute_huffman_tree(dstate *s, htree *t, const int *syms, const int *lens, int n, int which)
{
    int osym[MAXSYM], olen[MAXSYM], ocode[MAXSYM];
    int m = pzo_compute_code_values(syms, lens, n, osym, olen, ocode);
    int i, err = 0;
    tree_init(t);
    for (i = m - 1; i >= 0; i--) {
        int r = add_huffman_node(t, t->root, osym[i], olen[i], ocode[i], &err);
        if (r < 0) {
            char m2[160];
            switch (err) {
            case PZO_INS_TWO_VALUES:
                snprintf(m2, sizeof m2, "Huffman tree manipulation error: Two values point to the same place!");
                break;
            case PZO_INS_VALUE_HIT:
                snprintf(m2, sizeof m2, "Huffman tree manipulation error: HuffmanValue hit while inserting a value!");
                break;
            default:
                snprintf(m2, sizeof m2, "Huffman tree manipulation error: Tried to add where the leaf is a node: %d", osym[i]);
                break;
            }
            raise_err(s, PZO_E_HUFF_BUILD, (uint32_t)which | ((uint32_t)err << 8) | ((uint32_t)osym[i] << 16), 0, m2);
        }
        t->root = r;
    }
}

/* Monad.hs:295-302 nextCode over HuffmanTree.hs:73-83 advanceTree */
static int next_code(dstate *s, const htree *t)
{
    int node = t->root;
    for (;;) {
        uint32_t b = next_bits(s, 1);
        const hnode *nd = &t->n[node];
        int child;
        if (nd->kind == K_EMPTY)
            raise_err(s, PZO_E_HUFF_EMPTY_TREE, 0, 0,
                      "Huffman tree manipulation error: Tried to advance empty tree!");
        if (nd->kind == K_VALUE)
            raise_err(s, PZO_E_HUFF_ADVANCE_VALUE, 0, 0,
                      "Huffman tree manipulation error: Tried to advance value!");
        child = b ? nd->r : nd->l;
        if (t->n[child].kind == K_EMPTY)
            raise_err(s, PZO_E_HUFF_EMPTY_BRANCH, 0, 0,
                      "Huffman tree manipulation error: Advanced to empty tree!");
        if (t->n[child].kind == K_VALUE) return t->n[child].val;
        node = child;
    }
}

/* Deflate.hs:164-196 lengthArray: symbol 257..285 -> (base, extra bits) */
static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                      35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                      3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
/* Deflate.hs:203-237 distanceArray: code 0..29 -> (base, extra bits) */
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                       257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                       8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                       7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* Deflate.hs:290-292 codeLengthOrder */
static const int CODE_LENGTH_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* Deflate.hs:106-120 runInflate */
static void run_inflate(dstate *s, const htree *lit, const htree *dist)
{
    for (;;) {
        int code = next_code(s, lit);
        if (code < 256) {
            emit_byte(s, (uint8_t)code);
        } else if (code == 256) {
            return;
        } else {
            uint32_t len, d;
            int dcode;
            if (code > 285) { /* lengthArray ! c out of bounds (:160-166): the reference throws */
                char m[96];
                snprintf(m, sizeof m, "(reference throws) literal/length symbol %d has no length entry", code);
                raise_err(s, PZO_E_BAD_LITLEN_SYMBOL, (uint32_t)code, 0, m);
            }
            len = LEN_BASE[code - 257] + next_bits(s, LEN_EXTRA[code - 257]);
            dcode = next_code(s, dist);
            if (dcode > 29) { /* distanceArray ! c out of bounds (:199-205) */
                char m[96];
                snprintf(m, sizeof m, "(reference throws) distance symbol %d has no distance entry", dcode);
                raise_err(s, PZO_E_BAD_DIST_SYMBOL, (uint32_t)dcode, 0, m);
            }
            d = DIST_BASE[dcode] + next_bits(s, DIST_EXTRA[dcode]);
            emit_past_chunk(s, d, len);
            move_window(s); /* :119 */
        }
    }
}

/* Deflate.hs:124-156 getCodeLengths.  Returns the final n (may exceed maxl: overrun accepted). */
static int get_code_lengths(dstate *s, const htree *tree, int maxl, int *lens /* >= maxl+138 */)
{
    int n = 0, prev = 0, i;
    while (n < maxl) {
        int code = next_code(s, tree);
        int num, val;
        if (code <= 15) { /* :134-135 */
            lens[n++] = code;
            prev = code;
            continue;
        }
        if (code == 16) { /* :136-139 */
            num = 3 + (int)next_bits(s, 2);
            val = prev;
            if (n == 0) s->res->quirks |= PZO_QUIRK_REPEAT_NO_PREV;
            /* prev unchanged */
        } else if (code == 17) { /* :140-143 */
            num = 3 + (int)next_bits(s, 3);
            val = 0;
            prev = 0;
        } else if (code == 18) { /* :144-147 */
            num = 11 + (int)next_bits(s, 7);
            val = 0;
            prev = 0;
        } else { /* :148-149, unreachable: the code-length alphabet has symbols 0..18 only */
            char m[96];
            snprintf(m, sizeof m, "Decompression error: Unexpected code: %d", code);
            raise_err(s, PZO_E_BAD_CODELEN_SYMBOL, (uint32_t)code, 0, m);
            return 0;
        }
        if (n + num > maxl) s->res->quirks |= PZO_QUIRK_CODELEN_OVERRUN;
        for (i = 0; i < num; i++) lens[n + i] = val; /* addNTimes :150-156 */
        n += num;
    }
    return n;
}

/* Deflate.hs:65-104 inflateBlock */
static int inflate_block(dstate *s)
{
    int bfinal = (int)next_bits(s, 1); /* :67 */
    uint32_t btype = next_bits(s, 2);  /* :68 */
    s->res->n_blocks++;
    switch (btype) {
    case 0: { /* :70-78 */
        uint32_t len, nlen;
        advance_to_byte(s);
        len = next_word16(s);
        nlen = next_word16(s);
        if (len != ((~nlen) & 0xffffu))
            raise_err(s, PZO_E_FMT_LEN_NLEN, len, nlen,
                      "Block format error: Len/nlen mismatch in uncompressed block.");
        emit_stored(s, len);
        return bfinal;
    }
    case 1: /* :79-82 */
        run_inflate(s, &s->fixed_lit, &s->fixed_dist);
        return bfinal;
    case 2: { /* :83-101 */
        int hlit = 257 + (int)next_bits(s, 5);
        int hdist = 1 + (int)next_bits(s, 5);
        int hclen = 4 + (int)next_bits(s, 4);
        int cl_sym[19], cl_len[19];
        int lens[288 + 32 + 138 + 8];
        int syms[288 + 32 + 138 + 8];
        int i, n, ndist;
        for (i = 0; i < hclen; i++) { /* :87-88 */
            cl_len[i] = (int)next_bits(s, 3);
            cl_sym[i] = CODE_LENGTH_ORDER[i];
        }
        compute_huffman_tree(s, &s->code_tree, cl_sym, cl_len, hclen, PZO_TREE_CODELEN); /* :89 */
        memset(lens, 0, sizeof lens);
        n = get_code_lengths(s, &s->code_tree, hlit + hdist, lens); /* :90 */
        /* :94-97 split at hlit; everything at or past hlit (overrun included) is a distance symbol */
        for (i = 0; i < (int)(sizeof syms / sizeof syms[0]); i++) syms[i] = i;
        compute_huffman_tree(s, &s->lit_tree, syms, lens, hlit, PZO_TREE_LITLEN); /* :98 */
        ndist = n - hlit;
        compute_huffman_tree(s, &s->dist_tree, syms, lens + hlit, ndist, PZO_TREE_DIST); /* :99 */
        run_inflate(s, &s->lit_tree, &s->dist_tree); /* :100 */
        return bfinal;
    }
    default: /* :102-104 */
        raise_err(s, PZO_E_FMT_BTYPE, btype, 0, "Block format error: Unacceptable BTYPE: 3");
        return 1;
    }
}

/* Deflate.hs:241-251 buildFixedLitTree / buildFixedDistanceTree (rebuilt per stream, :41-42) */
static void build_fixed(dstate *s)
{
    int syms[288], lens[288], i;
    for (i = 0; i < 288; i++) {
        syms[i] = i;
        lens[i] = i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8;
    }
    compute_huffman_tree(s, &s->fixed_lit, syms, lens, 288, PZO_TREE_LITLEN);
    for (i = 0; i < 32; i++) lens[i] = 5;
    compute_huffman_tree(s, &s->fixed_dist, syms, lens, 32, PZO_TREE_DIST);
}

/* Zlib.hs:53-69 inflateWithHeaders + Deflate.hs:39-63 inflate/checkChecksum */
/* RFC 1952 member: header (2.3), deflate, CRC32 + ISIZE little-endian.  Extension, see PZO_F_GZIP. */
static void inflate_gzip_member_header(dstate *s)
{
    uint32_t hreg = 0xffffffffu, id1, id2, cm, flg, i;
    char m[128];
#define GZ_NEXT(dst) do { uint32_t b_ = next_byte(s); hreg = crc32_byte(hreg, (uint8_t)b_); (dst) = b_; } while (0)
    GZ_NEXT(id1);
    GZ_NEXT(id2);
    if (id1 != 0x1f || id2 != 0x8b) raise_err(s, PZO_E_GZIP_HEADER, 1, (id1 << 8) | id2, "Header error: gzip: bad magic");
    GZ_NEXT(cm);
    if (cm != 8) {
        snprintf(m, sizeof m, "Header error: gzip: bad compression method: %u", cm);
        raise_err(s, PZO_E_GZIP_HEADER, 2, cm, m);
    }
    GZ_NEXT(flg);
    if (flg & 0xe0) raise_err(s, PZO_E_GZIP_HEADER, 3, flg, "Header error: gzip: reserved flag bits set");
    for (i = 0; i < 6; i++) { uint32_t skip; GZ_NEXT(skip); (void)skip; } /* MTIME, XFL, OS */
    if (flg & 4) { /* FEXTRA */
        uint32_t lo, hi, xlen;
        GZ_NEXT(lo);
        GZ_NEXT(hi);
        xlen = lo | (hi << 8);
        for (i = 0; i < xlen; i++) { uint32_t skip; GZ_NEXT(skip); (void)skip; }
    }
    if (flg & 8) { uint32_t c; do { GZ_NEXT(c); } while (c != 0); }  /* FNAME */
    if (flg & 16) { uint32_t c; do { GZ_NEXT(c); } while (c != 0); } /* FCOMMENT */
    if (flg & 2) { /* FHCRC: the low 16 bits of the CRC-32 of the header so far */
        uint32_t want = ~hreg & 0xffffu, lo = next_byte(s), hi = next_byte(s), got = lo | (hi << 8);
        if (got != want) {
            snprintf(m, sizeof m, "Header error: gzip: header crc mismatch: %x != %x", got, want);
            raise_err(s, PZO_E_GZIP_HEADER, 4, got, m);
        }
    }
#undef GZ_NEXT
}

/* x^(8 n) mod P in the reflected representation, by square and multiply: the shift that crc32_combine applies */
static uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    int i;
    for (i = 0; i < 32; i++) {
        if ((a >> (31 - i)) & 1u) p ^= b;
        b = (b >> 1) ^ (0xedb88320u & (0u - (b & 1u)));
    }
    return p;
}
static uint32_t crc32_append(uint32_t crc_a, uint32_t crc_b, uint64_t len_b)
{
    uint32_t pw = 0x80000000u, sq = 0x00800000u;
    for (; len_b; len_b >>= 1) {
        if (len_b & 1u) pw = gf2_mul(pw, sq);
        sq = gf2_mul(sq, sq);
    }
    return gf2_mul(crc_a, pw) ^ crc_b;
}

/* RFC 1952 2.2: a gzip file is a series of members, decoded one after the other into one output (extension; pinned
 * against Python's gzip.decompress for valid files).  Every member's ISIZE is checked as it ends; the members'
 * CRC-32s are combined into the CRC the whole output must have, checked when decoding stops (and before a length
 * mismatch is reported, as zlib orders the two checks). */
static void inflate_gzip_members(dstate *s)
{
    uint32_t expect = 0;
    uint64_t mstart = 0;
    char m[128];
    s->crc = 0xffffffffu;
    for (;;) {
        uint32_t theirs, isize;
        int bad_len;
        inflate_gzip_member_header(s);
        build_fixed(s);
        for (;;) {
            int is_final = inflate_block(s);
            move_window(s);
            if (is_final) break;
        }
        advance_to_byte(s);
        theirs = next_word16(s);
        theirs |= next_word16(s) << 16;
        isize = next_word16(s);
        isize |= next_word16(s) << 16;
        expect = crc32_append(expect, theirs, s->total - mstart);
        bad_len = isize != (uint32_t)(s->total - mstart);
        if (bad_len || s->p + 2 > s->pend || s->p[0] != 0x1f || s->p[1] != 0x8b) { /* the last member (or a broken one) */
            uint32_t ours = ~s->crc;
            if (expect != ours) {
                snprintf(m, sizeof m, "Checksum error: checksum mismatch: %x != %x", expect, ours);
                raise_err(s, PZO_E_CHECKSUM, expect, ours, m);
            }
            if (bad_len) {
                snprintf(m, sizeof m, "Checksum error: gzip: length mismatch: %u != %u", isize, (uint32_t)(s->total - mstart));
                raise_err(s, PZO_E_GZIP_ISIZE, isize, (uint32_t)(s->total - mstart), m);
            }
            return;
        }
        mstart = s->total;
    }
}

static void inflate_with_headers(dstate *s)
{
    uint32_t cmf = next_byte(s);
    uint32_t flg = next_byte(s);
    uint32_t both = (cmf << 8) | flg;
    uint32_t cm = cmf & 0x0f, cinfo = cmf >> 4;
    char m[96];
    if (both % 31 != 0) /* :62-63 */
        raise_err(s, PZO_E_HDR_FCHECK, both, 0, "Header error: Header checksum failed");
    if (cm != 8) { /* :64-65 */
        snprintf(m, sizeof m, "Header error: Bad compression method: %u", cm);
        raise_err(s, PZO_E_HDR_METHOD, cm, 0, m);
    }
    if (cinfo > 7) { /* :66-67 */
        snprintf(m, sizeof m, "Header error: Window size too big: %u", cinfo);
        raise_err(s, PZO_E_HDR_WINDOW, cinfo, 0, m);
    }
    if (flg & 0x20) { /* :68 skip DICTID, no dictionary is installed */
        uint32_t id = 0;
        int i;
        for (i = 0; i < 4; i++) id = (id << 8) | next_byte(s);
        if (s->dict_len == 0) {
            s->res->quirks |= PZO_QUIRK_FDICT_SKIPPED;
        } else { /* EXTENSION (not in the reference): RFC 1950 2.2 -- DICTID is the Adler-32 of the dictionary */
            uint32_t ours = pzo_adler32(1, s->dict, s->dict_len);
            if (id != ours) {
                snprintf(m, sizeof m, "Header error: preset dictionary mismatch: %x != %x", id, ours);
                s->dict_len = 0;
                raise_err(s, PZO_E_DICT, id, ours, m);
            }
        }
    } else {
        s->dict_len = 0; /* a dictionary is history only for a stream that asks for one */
    }
    build_fixed(s); /* Deflate.hs:41-42 */
    for (;;) {      /* Deflate.hs:45-50 go */
        int is_final = inflate_block(s);
        move_window(s);
        if (is_final) break;
    }
    { /* Deflate.hs:52-63 checkChecksum */
        uint32_t ours, theirs;
        advance_to_byte(s);
        ours = (s->b << 16) | s->a;
        theirs = next_word32(s);
        if (theirs != ours) {
            snprintf(m, sizeof m, "Checksum error: checksum mismatch: %x != %x", theirs, ours);
            raise_err(s, PZO_E_CHECKSUM, theirs, ours, m);
        }
    }
}

static int decompress_core(const uint8_t *in, const uint64_t *chunk_off, uint32_t n_chunks,
                           uint8_t *out, uint64_t out_cap, uint32_t flags, pzo_result *res,
                           const uint8_t *dict, uint64_t dict_len,
                           int32_t *ev_type, uint32_t *ev_val, uint32_t ev_cap, uint32_t *n_events)
{
    static __thread dstate st; /* large (tries + window): keep it off the stack */
    dstate *s = &st;
    memset(res, 0, sizeof *res);
    s->ev_type = ev_type;
    s->ev_val = ev_val;
    s->ev_cap = ev_cap;
    s->ev_n = 0;
    s->dict = dict;
    s->dict_len = dict ? dict_len : 0;
    s->base = in;
    s->coff = chunk_off;
    s->nchunks = n_chunks;
    s->chunk_next = 0;
    s->p = s->pend = in;
    s->bitno = 8; /* Monad.hs:172-179 initialState */
    s->cur = 0;
    s->a = 1;
    s->b = 0;
    s->out = out;
    s->cap = out ? out_cap : 0;
    s->total = 0;
    s->ow_next = 0;
    s->flags = flags;
    s->res = res;
    s->crc = 0xffffffffu;
    if (setjmp(s->jb) == 0) {
        if (flags & PZO_F_GZIP) inflate_gzip_members(s);
        else inflate_with_headers(s);
        /* Zlib.hs:46-49: Done with chunks left over is an error, Done with none is Right */
        if (s->chunk_next < s->nchunks) {
            res->status = PZO_E_DATA_REMAINING;
            snprintf(res->message, sizeof res->message,
                     "Decompression error: Finished with data remaining.");
        } else if (s->total > s->cap) {
            res->status = PZO_E_OUT_TOO_SMALL;
            snprintf(res->message, sizeof res->message, "(not a reference outcome) output buffer too small");
        }
    }
    /* the incremental protocol's last events: finalize (Monad.hs:349-353) publishes what is left in the window, then
     * Done; or the DecompError.  ("Finished with data remaining." belongs to decompress, Zlib.hs:46-49, not to it.) */
    if (res->status == PZO_OK || res->status == PZO_E_DATA_REMAINING || res->status == PZO_E_OUT_TOO_SMALL) {
        trace_event(s, PZO_EV_CHUNK, (uint32_t)s->ow_next);
        trace_event(s, PZO_EV_DONE, 0);
    } else if (res->status != PZO_E_TRUNCATED) {
        trace_event(s, PZO_EV_ERROR, (uint32_t)res->status);
    }
    if (n_events) *n_events = s->ev_n;
    res->adler = (flags & PZO_F_GZIP) ? ~s->crc : ((s->b << 16) | s->a);
    res->out_len = s->total;
    res->in_used = (uint64_t)(s->p - in);
    return res->status;
}

int pzo_decompress_chunks(const uint8_t *in, const uint64_t *chunk_off, uint32_t n_chunks,
                          uint8_t *out, uint64_t out_cap, uint32_t flags, pzo_result *res)
{
    return decompress_core(in, chunk_off, n_chunks, out, out_cap, flags, res, NULL, 0, NULL, NULL, 0, NULL);
}

/* decompressIncremental (Zlib.hs, Monad.hs:163-197) driven the way Deflate.hs:30-48 drives it, one input piece per
 * NeedMore: the sequence of ZlibDecoder constructors it goes through.  ev_val = chunk length / status. */
int pzo_trace(const uint8_t *in, const uint64_t *chunk_off, uint32_t n_chunks, uint8_t *out, uint64_t out_cap,
              int32_t *ev_type, uint32_t *ev_val, uint32_t ev_cap, uint32_t *n_events, pzo_result *res)
{
    return decompress_core(in, chunk_off, n_chunks, out, out_cap, 0, res, NULL, 0, ev_type, ev_val, ev_cap, n_events);
}

/* EXTENSION (the reference skips DICTID, Zlib.hs:68): one chunk, with the preset dictionary the stream was made with */
int pzo_decompress_dict(const uint8_t *in, uint64_t in_len, const uint8_t *dict, uint64_t dict_len,
                        uint8_t *out, uint64_t out_cap, pzo_result *res)
{
    uint64_t off[2];
    off[0] = 0;
    off[1] = in_len;
    return decompress_core(in, off, in_len ? 1u : 0u, out, out_cap, 0, res, dict, dict_len, NULL, NULL, 0, NULL);
}

int pzo_decompress(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap, pzo_result *res)
{
    /* L.fromStrict: one chunk, or none when empty (an empty lazy ByteString has no chunks) */
    uint64_t off[2];
    off[0] = 0;
    off[1] = in_len;
    return pzo_decompress_chunks(in, off, in_len ? 1u : 0u, out, out_cap, 0, res);
}

uint32_t pzo_decompress_many(const uint8_t *in_base, const uint64_t *in_off,
                             uint8_t *out_base, const uint64_t *out_off,
                             uint64_t *out_len, int32_t *status, uint32_t *adler, uint32_t n)
{
    uint32_t i, bad = 0;
    for (i = 0; i < n; i++) {
        pzo_result r;
        pzo_decompress(in_base + in_off[i], in_off[i + 1] - in_off[i],
                       out_base + out_off[i], out_off[i + 1] - out_off[i], &r);
        if (out_len) out_len[i] = r.out_len;
        if (status) status[i] = r.status;
        if (adler) adler[i] = r.adler;
        if (r.status != PZO_OK) bad++;
    }
    return bad;
}
