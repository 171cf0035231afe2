// inflate_core.h -- one zlib stream decoded by one wavefront.
//
// MI355X counterpart of pure-zlib's whole decode stack:
//   Zlib.hs:53-69      inflateWithHeaders   -> inflate_stream() prologue
//   Deflate.hs:39-63   inflate/checkChecksum-> inflate_stream() block loop + trailer
//   Deflate.hs:65-104  inflateBlock         -> stored_block() / dynamic_header() / fixed tables
//   Deflate.hs:106-120 runInflate           -> token_loop()
//   Deflate.hs:124-156 getCodeLengths       -> dynamic_header()
//   Deflate.hs:160-237 length/distance arrays -> litlen_entry()/dist_entry() (closed forms)
//   Deflate.hs:255-292 computeCodeValues    -> build_table() (canonical codes, wave-parallel)
//   HuffmanTree.hs     binary trie          -> two-level LDS table: a direct 2^P LUT indexed by the
//                                             next P stream bits, then a canonical first-code/count
//                                             table + symbol permutation for codes longer than P
//   Monad.hs:203-307   bit/byte reader      -> BitReader (coalesced dword loads, 64-bit wave-uniform
//                                             bit buffer in SGPRs, per-wave bit cursor)
//   OutputWindow.hs    128 KiB flat window  -> 2^RING_BITS LDS ring, lane-cooperative LZ77 copy
//   Adler32.hs         per-byte checksum    -> folded into the ring->HBM flush as a wave reduction
//
// All decode state is wave-uniform; the 64 lanes cooperate on table construction, match
// copies, stored-block copies and the flush.  The same source compiles as a one-lane host
// program for the CPU model tests (see wave.h).
#pragma once
#include <stdint.h>

#include "wave.h"

namespace pzg {

// ---- per-stream status codes: numerically identical to include/pzg.h -----------------------
enum : int32_t {
    ST_OK = 0,
    ST_TRUNCATED = 1,
    ST_HDR_FCHECK = 2,
    ST_HDR_METHOD = 3,
    ST_HDR_WINDOW = 4,
    ST_FMT_LEN_NLEN = 5,
    ST_FMT_BTYPE = 6,
    ST_HUFF_BUILD = 7,
    ST_HUFF_EMPTY_TREE = 8,
    ST_HUFF_EMPTY_BRANCH = 9,
    ST_CHECKSUM = 10,
    ST_BAD_DISTANCE = 11,
    ST_BAD_LITLEN_SYMBOL = 12,
    ST_BAD_DIST_SYMBOL = 13,
    ST_OUT_TOO_SMALL = 14
};

enum { TREE_CODELEN = 0, TREE_LITLEN = 1, TREE_DIST = 2 };

constexpr int LIT_BITS = 10;  // primary literal/length LUT: 2^10 x 4 B = 4 KiB
constexpr int DIST_BITS = 8;  // primary distance LUT:        2^8  x 4 B = 1 KiB
constexpr int CL_BITS = 7;    // code-length code: max length 7, the LUT is exhaustive
constexpr uint32_t ADLER_MOD = 65521u;

constexpr int MAX_LIT_SYMS = 288;        // HLIT <= 288
constexpr int MAX_DIST_SYMS = 32 + 138;  // HDIST <= 32 plus a code-length repeat overrun (Deflate.hs:132)
constexpr int MAX_LENS = 288 + 32 + 138;

// LUT entry: [4:0] code bits n  [8:5] extra bits e  [11:9] kind  [31:16] value
enum : uint32_t {
    K_LIT = 0,           // value = literal byte (or code-length symbol)
    K_BASE = 1,          // value = base length / base distance, e extra bits follow
    K_EOB = 2,           // symbol 256
    K_LONG = 3,          // code longer than the primary table: second level
    K_EMPTY_BRANCH = 4,  // n = depth at which the reference's walk reaches HuffmanEmpty
    K_EMPTY_TREE = 5,    // the tree has no codes at all
    K_BADSYM = 6         // symbol 286/287 or distance symbol >= 30: value = symbol
};

PZG_FN uint32_t mk_entry(uint32_t n, uint32_t e, uint32_t kind, uint32_t value)
{
    return n | (e << 5) | (kind << 9) | (value << 16);
}
PZG_FN uint32_t ent_n(uint32_t x) { return x & 31u; }
PZG_FN uint32_t ent_e(uint32_t x) { return (x >> 5) & 15u; }
PZG_FN uint32_t ent_kind(uint32_t x) { return (x >> 9) & 7u; }
PZG_FN uint32_t ent_val(uint32_t x) { return x >> 16; }

// Deflate.hs:164-196 lengthArray as a closed form: symbol 257..285 -> (base, extra)
PZG_FN uint32_t litlen_entry(uint32_t sym, uint32_t n)
{
    if (sym < 256u) return mk_entry(n, 0, K_LIT, sym);
    if (sym == 256u) return mk_entry(n, 0, K_EOB, 0);
    if (sym > 285u) return mk_entry(n, 0, K_BADSYM, sym);
    uint32_t i = sym - 257u;
    if (i < 8u) return mk_entry(n, 0, K_BASE, 3u + i);
    if (i == 28u) return mk_entry(n, 0, K_BASE, 258u);
    uint32_t e = (i >> 2) - 1u;
    return mk_entry(n, e, K_BASE, 3u + ((4u + (i & 3u)) << e));
}

// Deflate.hs:203-237 distanceArray as a closed form: code 0..29 -> (base, extra)
PZG_FN uint32_t dist_entry(uint32_t sym, uint32_t n)
{
    if (sym > 29u) return mk_entry(n, 0, K_BADSYM, sym);
    if (sym < 4u) return mk_entry(n, 0, K_BASE, 1u + sym);
    uint32_t e = (sym >> 1) - 1u;
    return mk_entry(n, e, K_BASE, 1u + ((2u + (sym & 1u)) << e));
}

// code-length alphabet (Deflate.hs:131-149): 0..15 literal lengths, 16/17/18 repeats with 2/3/7 extra bits
PZG_FN uint32_t codelen_entry(uint32_t sym, uint32_t n)
{
    uint32_t e = sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
    return mk_entry(n, e, K_LIT, sym);
}

// ---- LDS image of one wave ------------------------------------------------------------------
struct TreeMeta {          // second-level (canonical) decode tables, index = code length 1..15
    uint16_t count[16];    // symbols of that length
    uint16_t first[16];    // first canonical code of that length
    uint16_t offs[16];     // index of its first symbol in the sorted permutation
};

template <int RING_BITS>
struct alignas(16) WaveLds {
    uint8_t ring[1u << RING_BITS];       // OutputWindow: the last 2^RING_BITS bytes produced
    uint32_t lit_lut[1u << LIT_BITS];    // HuffmanTree (literal/length), level 1
    uint32_t dist_lut[1u << DIST_BITS];  // HuffmanTree (distance), level 1; the code-length LUT while a header is read
    uint16_t lit_sorted[MAX_LIT_SYMS];   // level 2: symbols in canonical order
    uint16_t dist_sorted[MAX_DIST_SYMS + 6];
    TreeMeta lit_meta;
    TreeMeta dist_meta;
    uint32_t cnt[16];                    // histogram scratch for build_table
    uint8_t lens[MAX_LENS + 22];         // code lengths of the block being set up
    uint8_t cl_lens[20];                 // code-length code lengths in symbol order
};

// ---- result of one stream -------------------------------------------------------------------
struct StreamResult {
    int32_t status;
    uint32_t detail0, detail1;
    uint32_t adler;
    uint64_t out_len;
    uint64_t in_used;
};

// ---- Monad.hs:203-307: the bit reader ---------------------------------------------------------
// Each lane holds one dword of the current PZG_WAVE-dword input chunk (one coalesced load per
// chunk, the next chunk prefetched); the wave-uniform 64-bit bit buffer is fed from it with
// v_readlane.  Bits are consumed LSB-first (Monad.hs:224-230).
struct BitReader {
    const uint32_t *base;  // 4-byte aligned address at or below the stream start
    uint32_t ndw;          // dwords covering [base, stream end)
    uint32_t mis_bits;     // 8 * (stream start - base)
    uint64_t end_rel;      // mis_bits + 8 * stream length: first bit (relative to base) past the stream
    uint64_t buf;          // wave-uniform
    uint32_t cnt;          // valid bits in buf
    uint32_t next;         // next dword index to feed
    uint32_t cur, nxt;     // per-lane: dwords of the current / next chunk

    PZG_FN uint32_t load_dw(uint32_t i) const { return i < ndw ? base[i] : 0u; }

    PZG_FN void start(const uint8_t *in, uint64_t in_len, uint64_t byte_pos)
    {
        const uint8_t *p = in + byte_pos;
        uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
        uint64_t remain = in_len - byte_pos;
        base = (const uint32_t *)(p - mis);
        mis_bits = mis * 8u;
        ndw = (uint32_t)((mis + remain + 3u) >> 2);
        end_rel = (uint64_t)mis_bits + remain * 8u;
        uint32_t l = lane_id();
        cur = load_dw(l);
        nxt = load_dw(PZG_WAVE + l);
        next = 0;
        buf = 0;
        cnt = 0;
        refill();
        buf >>= mis_bits;  // cnt >= 33 > 24 >= mis_bits
        cnt -= mis_bits;
    }

    PZG_FN void refill()
    {
        while (cnt <= 32u) {
            uint32_t d = read_lane(cur, next & (PZG_WAVE - 1u));
            buf |= (uint64_t)d << cnt;
            cnt += 32u;
            next++;
            if ((next & (PZG_WAVE - 1u)) == 0u) {
                cur = nxt;
                nxt = load_dw(next + PZG_WAVE + lane_id());
            }
        }
    }

    // bit position of the next unread bit, relative to `base`
    PZG_FN uint64_t pos_rel() const { return (uint64_t)next * 32u - cnt; }
    // real (in-stream) bits still unread; <= 0 means everything in buf is padding
    PZG_FN int64_t avail() const { return (int64_t)end_rel - (int64_t)pos_rel(); }
    PZG_FN uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    PZG_FN void drop(uint32_t n)
    {
        buf >>= n;
        cnt -= n;
    }
};

// ---- decoder state (all wave-uniform) -----------------------------------------------------------
template <int RING_BITS>
struct Decoder {
    static constexpr uint32_t RING = 1u << RING_BITS;
    static constexpr uint32_t RMASK = RING - 1u;
    static constexpr uint32_t FLUSH_AT = RING - 512u;

    WaveLds<RING_BITS> &L;
    const uint8_t *in;
    uint64_t in_len;
    uint8_t *out;
    uint64_t cap;
    BitReader br;
    uint64_t in_byte0;  // byte offset of br.base's stream start (br was started at this byte position)
    uint64_t op;        // bytes produced
    uint64_t flushed;   // bytes already written to HBM and folded into the Adler state
    uint32_t adler_a, adler_b;
    uint32_t lit_e15, dist_e15;  // Kraft totals in 2^-15 units (0 = empty tree)
    int fixed_loaded;            // lit/dist tables currently hold the fixed code
    int32_t status;
    uint32_t detail0, detail1;

    PZG_FN Decoder(WaveLds<RING_BITS> &lds) : L(lds) {}

    PZG_FN int fail(int32_t st, uint32_t d0, uint32_t d1)
    {
        status = st;
        detail0 = d0;
        detail1 = d1;
        return st;
    }

    // absolute bit offset of the next unread bit within the stream
    PZG_FN uint64_t stream_bit_pos() const { return in_byte0 * 8u + br.pos_rel() - br.mis_bits; }

    // ---- OutputWindow.hs + Adler32.hs: ring -> HBM flush with the checksum folded in ------------
    // Writes produced bytes [flushed, to) and advances the Adler state over them.  `flushed` is
    // always a multiple of 16, so ring offsets and (for a 16-byte aligned output) global
    // addresses are 16-byte aligned: one ds_read_b128 + one global_store_dwordx4 per lane.
    PZG_FN void flush_to(uint64_t to)
    {
        wave_sync();
        const uint64_t from = flushed;
        if (to <= from) return;
        const uint32_t n = (uint32_t)(to - from);
        const uint32_t nvec = (n + 15u) >> 4;
        const uint32_t lane = lane_id();
        const bool out_al = (((uintptr_t)out) & 15u) == 0u;
        uint32_t a_l = 0, w_l = 0, u_l = 0;
        for (uint32_t it = 0; it * PZG_WAVE < nvec; ++it) {
            const uint32_t j = it * PZG_WAVE + lane;
            if (j < nvec) {
                const uint64_t pos = from + (uint64_t)j * 16u;
                const uint32_t roff = (uint32_t)pos & RMASK;
#if PZG_DEVICE_PASS
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 rv = *(const u32x4 *)(const void *)&L.ring[roff];  // ds_read_b128
                uint32_t x0 = rv.x, x1 = rv.y, x2 = rv.z, x3 = rv.w;
#else
                const uint32_t *rp = (const uint32_t *)(const void *)&L.ring[roff];
                uint32_t x0 = rp[0], x1 = rp[1], x2 = rp[2], x3 = rp[3];
#endif
                const uint32_t valid = (to - pos) >= 16u ? 16u : (uint32_t)(to - pos);
                if (valid < 16u) {  // zero the bytes past `to`: they add nothing to either sum
                    uint32_t m0 = valid >= 4u ? ~0u : ((1u << (8u * valid)) - 1u);
                    uint32_t m1 = valid >= 8u ? ~0u : valid <= 4u ? 0u : ((1u << (8u * (valid - 4u))) - 1u);
                    uint32_t m2 = valid >= 12u ? ~0u : valid <= 8u ? 0u : ((1u << (8u * (valid - 8u))) - 1u);
                    uint32_t m3 = valid <= 12u ? 0u : ((1u << (8u * (valid - 12u))) - 1u);
                    x0 &= m0;
                    x1 &= m1;
                    x2 &= m2;
                    x3 &= m3;
                }
                if (out_al && valid == 16u && pos + 16u <= cap) {
                    uint32_t *gp = (uint32_t *)(void *)(out + pos);
#if PZG_DEVICE_PASS
                    u32x4 v = {x0, x1, x2, x3};
                    *(u32x4 *)gp = v;
#else
                    gp[0] = x0; gp[1] = x1; gp[2] = x2; gp[3] = x3;
#endif
                } else {
                    const uint32_t xs[4] = {x0, x1, x2, x3};
                    for (uint32_t k = 0; k < valid; ++k)
                        if (pos + k < cap) out[pos + k] = (uint8_t)(xs[k >> 2] >> (8u * (k & 3u)));
                }
                // Adler32.hs:29-34 advanceNoMod over 16 bytes at once: s = sum d_i, t = sum (16-i) d_i
                uint32_t s = sum4(x0, sum4(x1, sum4(x2, sum4(x3, 0u))));
                uint32_t t = dot4(x0, 0x0D0E0F10u, dot4(x1, 0x090A0B0Cu, dot4(x2, 0x05060708u, dot4(x3, 0x01020304u, 0u))));
                a_l += s;
                w_l += t;
                u_l += it * s;
            }
        }
        // weight of byte i of vector j = it*WAVE + lane is  n - 16 j - i
        //   = (n - 16 lane - 16) - 16*WAVE*it + (16 - i)
        int64_t bl = (int64_t)((int64_t)n - 16 * (int64_t)lane - 16) * (int64_t)a_l + (int64_t)w_l -
                     (int64_t)(16u * PZG_WAVE) * (int64_t)u_l;
        uint32_t bl_mod = (uint32_t)((uint64_t)bl % ADLER_MOD);
        uint32_t sum_a = wave_sum(a_l);
        uint32_t sum_b = wave_sum(bl_mod);
        // Adler32.hs:22-27 in block form: A' = A + sum d ; B' = B + n*A + sum (n - pos) d
        uint64_t nb = (uint64_t)adler_b + (uint64_t)(n % ADLER_MOD) * adler_a + sum_b;
        adler_a = (uint32_t)(((uint64_t)adler_a + sum_a) % ADLER_MOD);
        adler_b = (uint32_t)(nb % ADLER_MOD);
        flushed = to;
        wave_sync();
    }

    PZG_FN void maybe_flush()
    {
        if (op - flushed >= FLUSH_AT) flush_to(op & ~(uint64_t)15u);
    }

    // Monad.hs:309-315 emitByte -> OutputWindow.hs:64-68 addByte
    PZG_FN void put_literal(uint32_t v)
    {
        if (lane_id() == 0u) L.ring[(uint32_t)op & RMASK] = (uint8_t)v;
        op++;
    }

    // Monad.hs:324-333 emitPastChunk -> OutputWindow.hs:82-101 addOldChunk/copyChunked.
    // Lane-cooperative LZ77 copy.  Every source byte lies before `op`: for dist >= len it is
    // op-dist+k, for dist < len the pattern repeats with period dist (copyChunked's dist-sized
    // pieces), i.e. op-dist+(k mod dist).  So all reads are issued before any write.
    PZG_FN void copy_match(uint32_t dist, uint32_t len)
    {
        constexpr uint32_t MAXCH = (258u + PZG_WAVE - 1u) / PZG_WAVE;
        const uint32_t lane = lane_id();
        const uint32_t src0 = (uint32_t)op - dist;
        const uint32_t dst0 = (uint32_t)op;
        uint8_t v[MAXCH];
        const bool overlap = dist < len;
#if PZG_DEVICE_PASS
        const float rd = overlap ? __builtin_amdgcn_rcpf((float)dist) : 0.0f;
#endif
#pragma unroll
        for (uint32_t c = 0; c < MAXCH; ++c) {
            const uint32_t k = c * PZG_WAVE + lane;
            if (c * PZG_WAVE < len && k < len) {
                uint32_t off = k;
                if (overlap) {
#if PZG_DEVICE_PASS
                    uint32_t q = (uint32_t)(((float)k + 0.5f) * rd);
                    off = k - q * dist;
                    if (off >= dist) off -= dist;
#else
                    off = k % dist;
#endif
                }
                v[c] = L.ring[(src0 + off) & RMASK];
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < MAXCH; ++c) {
            const uint32_t k = c * PZG_WAVE + lane;
            if (c * PZG_WAVE < len && k < len) L.ring[(dst0 + k) & RMASK] = v[c];
        }
        op += len;
    }

    // ---- Deflate.hs:255-292 + HuffmanTree.hs: canonical code -> two-level table -----------------
    // lens[0..n) in LDS.  Builds the 2^P primary LUT, the sorted-symbol permutation and the
    // per-length first/count/offset table.  Returns false when the code is over-subscribed,
    // which is exactly when createHuffmanTree returns Left (any overlap of canonical codes).
    template <int P, int TREE>
    PZG_FN bool build_table(const uint8_t *lens, uint32_t n, uint32_t *lut, uint16_t *sorted, TreeMeta *meta,
                            uint32_t *e15_out)
    {
        const uint32_t lane = lane_id();
        // pass 1: histogram of code lengths (blCount, Deflate.hs:266)
        wave_sync();
        if (lane < 16u || PZG_WAVE == 1u)
            for (uint32_t i = lane; i < 16u; i += PZG_WAVE) L.cnt[i] = 0u;
        wave_sync();
        for (uint32_t s = lane; s < n; s += PZG_WAVE) {
            uint32_t len = lens[s];
#if PZG_DEVICE_PASS
            if (len) atomicAdd(&L.cnt[len], 1u);
#else
            if (len) L.cnt[len]++;
#endif
        }
        wave_sync();
        // next_code (step2, Deflate.hs:273-278), offsets, Kraft sum; all wave-uniform
        uint32_t count[16], first[16], offs[16];
        uint32_t code = 0, off = 0, e15 = 0, maxlen = 0;
        count[0] = 0;
        first[0] = 0;
        offs[0] = 0;
#pragma unroll
        for (uint32_t l = 1; l < 16u; ++l) {
            uint32_t c = uni(L.cnt[l]);
            count[l] = c;
            code = (code + count[l - 1]) << 1;
            first[l] = code;
            offs[l] = off;
            off += c;
            e15 += c << (15u - l);
            if (c) maxlen = l;
        }
        *e15_out = e15;
        if (e15 > 32768u) return false;  // over-subscribed: some insertion must collide
        if (meta) {
            for (uint32_t l = lane; l < 16u; l += PZG_WAVE) {
                // per-lane select from the uniform arrays without dynamic register indexing
                uint32_t c = 0, f = 0, o = 0;
#pragma unroll
                for (uint32_t q = 0; q < 16u; ++q) {
                    if (q == l) {
                        c = count[q];
                        f = first[q];
                        o = offs[q];
                    }
                }
                meta->count[l] = (uint16_t)c;
                meta->first[l] = (uint16_t)f;
                meta->offs[l] = (uint16_t)o;
            }
        }
        // default fill: patterns no code of length <= P covers are either the prefix of a longer
        // code (K_LONG) or lead the reference's trie walk into HuffmanEmpty at some depth d
        // (HuffmanTree.hs:78-80): the first d whose d-bit prefix lies at or past the end of all codes.
        // covered_p: number of P-bit prefixes covered by codes of length <= P (they are [0, covered_p))
        (void)maxlen;
        const uint32_t covered_p = first[P] + count[P];
        if (covered_p < (1u << P)) {
            for (uint32_t idx = lane; idx < (1u << P); idx += PZG_WAVE) {
                uint32_t c_p = bitrev32(idx) >> (32u - P);  // MSB-first value of the P stream bits
                if (c_p < covered_p) continue;
                uint32_t ent;
                if (e15 == 0u) {
                    ent = mk_entry(1, 0, K_EMPTY_TREE, 0);
                } else if ((c_p << (15u - P)) < e15) {
                    ent = mk_entry(0, 0, K_LONG, 0);
                } else {
                    uint32_t d = 1;
                    while (d < (uint32_t)P && (((c_p >> ((uint32_t)P - d)) << (15u - d)) < e15)) d++;
                    ent = mk_entry(d, 0, K_EMPTY_BRANCH, 0);
                }
                lut[idx] = ent;
            }
        }
        wave_sync();
        // pass 2: canonical code of every symbol (step3, Deflate.hs:280-288): first[len] + rank among
        // the symbols of equal length below it; then the replicated LUT fill.
        uint32_t basec[16];
#pragma unroll
        for (uint32_t l = 0; l < 16u; ++l) basec[l] = 0;
        for (uint32_t s0 = 0; s0 < n; s0 += PZG_WAVE) {
            const uint32_t s = s0 + lane;
            const uint32_t len = s < n ? lens[s] : 0u;
            uint32_t rank = 0, fcode = 0, soff = 0;
#pragma unroll
            for (uint32_t l = 1; l < 16u; ++l) {
                if (count[l] == 0u) continue;  // uniform
                const bool mine = len == l;
                const uint64_t m = ballot(mine);
                if (mine) {
                    rank = basec[l] + mbcnt(m);
                    fcode = first[l];
                    soff = offs[l];
                }
                basec[l] += popc64(m);
            }
            if (len != 0u) {
                const uint32_t c = fcode + rank;
                if (sorted) sorted[soff + rank] = (uint16_t)s;
                if (len <= (uint32_t)P) {
                    uint32_t ent = TREE == TREE_LITLEN ? litlen_entry(s, len)
                                   : TREE == TREE_DIST ? dist_entry(s, len)
                                                       : codelen_entry(s, len);
                    const uint32_t rev = bitrev32(c) >> (32u - len);
                    for (uint32_t idx = rev; idx < (1u << P); idx += (1u << len)) lut[idx] = ent;
                }
            }
        }
        wave_sync();
        return true;
    }

    // Second level (codes longer than the primary table): canonical first-code walk, one length
    // per step, equivalent to HuffmanTree.hs:73-83 advanceTree on the same bits.  Returns an entry.
    template <int TREE>
    PZG_FN uint32_t decode_long(const TreeMeta *meta, const uint16_t *sorted, uint32_t e15)
    {
        uint32_t code = 0;
        uint32_t bits = (uint32_t)br.buf;
        for (uint32_t l = 1; l < 16u; ++l) {
            code = (code << 1) | (bits & 1u);
            bits >>= 1;
            uint32_t cnt_l = uni(meta->count[l]);
            uint32_t first_l = uni(meta->first[l]);
            if (code - first_l < cnt_l && code >= first_l) {
                uint32_t sym = uni(sorted[uni(meta->offs[l]) + code - first_l]);
                return TREE == TREE_LITLEN ? litlen_entry(sym, l) : dist_entry(sym, l);
            }
            if ((code << (15u - l)) >= e15) return mk_entry(l, 0, K_EMPTY_BRANCH, 0);
        }
        return mk_entry(15, 0, K_EMPTY_BRANCH, 0);  // unreachable: e15 <= 2^15 ends every walk by 15
    }

    // Checks a decoded entry against the real bits left, in the reference's order: the walk/extra
    // bits run out of data (TRUNCATED) before any error that needs a later bit.
    PZG_FN int check_entry(uint32_t ent, int32_t empty_branch_status)
    {
        (void)empty_branch_status;
        const uint32_t kind = ent_kind(ent);
        const int64_t av = br.avail();
        if (kind == K_EMPTY_TREE) {
            // nextCode reads one bit, then advanceTree fails (Monad.hs:297-299, HuffmanTree.hs:76)
            if (av < 1) return fail(ST_TRUNCATED, 0, 0);
            return fail(ST_HUFF_EMPTY_TREE, 0, 0);
        }
        if (kind == K_EMPTY_BRANCH) {
            if (av < (int64_t)ent_n(ent)) return fail(ST_TRUNCATED, 0, 0);
            return fail(ST_HUFF_EMPTY_BRANCH, 0, 0);
        }
        if (av < (int64_t)ent_n(ent)) return fail(ST_TRUNCATED, 0, 0);
        return ST_OK;
    }

    // ---- Deflate.hs:106-120 runInflate ------------------------------------------------------------
    PZG_FN int token_loop()
    {
        for (;;) {
            br.refill();
            uint32_t e = uni(L.lit_lut[br.peek(LIT_BITS)]);
            uint32_t kind = ent_kind(e);
            if (kind == K_LONG) {
                e = decode_long<TREE_LITLEN>(&L.lit_meta, L.lit_sorted, lit_e15);
                kind = ent_kind(e);
            }
            const bool tail = br.pos_rel() + 64u > br.end_rel;  // padding bits may be in the buffer
            if (kind == K_LIT) {
                const uint32_t n = ent_n(e);
                if (tail && br.avail() < (int64_t)n) return fail(ST_TRUNCATED, 0, 0);
                br.drop(n);
                put_literal(ent_val(e));
                maybe_flush();
                continue;
            }
            if (kind == K_BASE) {
                const uint32_t n = ent_n(e), ex = ent_e(e);
                if (tail && br.avail() < (int64_t)(n + ex)) return fail(ST_TRUNCATED, 0, 0);
                const uint32_t len = ent_val(e) + (((uint32_t)(br.buf >> n)) & ((1u << ex) - 1u));
                br.drop(n + ex);
                br.refill();
                uint32_t d = uni(L.dist_lut[br.peek(DIST_BITS)]);
                uint32_t dk = ent_kind(d);
                if (dk == K_LONG) {
                    d = decode_long<TREE_DIST>(&L.dist_meta, L.dist_sorted, dist_e15);
                    dk = ent_kind(d);
                }
                if (dk != K_BASE) {
                    if (int st = check_entry(d, 0)) return st;
                    // K_BADSYM: distanceArray ! c out of range, the reference throws (Deflate.hs:199-205)
                    return fail(ST_BAD_DIST_SYMBOL, ent_val(d), 0);
                }
                const uint32_t dn = ent_n(d), dex = ent_e(d);
                if (tail && br.avail() < (int64_t)(dn + dex)) return fail(ST_TRUNCATED, 0, 0);
                const uint32_t dist = ent_val(d) + (((uint32_t)(br.buf >> dn)) & ((1u << dex) - 1u));
                br.drop(dn + dex);
                if ((uint64_t)dist > op) return fail(ST_BAD_DISTANCE, dist, (uint32_t)op);
                copy_match(dist, len);
                maybe_flush();
                continue;
            }
            if (int st = check_entry(e, 0)) return st;
            if (kind == K_EOB) {
                br.drop(ent_n(e));
                return ST_OK;
            }
            // K_BADSYM: lengthArray ! c out of range, the reference throws (Deflate.hs:160-166)
            return fail(ST_BAD_LITLEN_SYMBOL, ent_val(e), 0);
        }
    }

    // ---- Deflate.hs:70-78: stored block -------------------------------------------------------------
    PZG_FN int stored_block()
    {
        br.drop(br.cnt & 7u);  // advanceToByte (Monad.hs:304-307); buf always ends on a byte boundary
        br.refill();
        if (br.avail() < 32) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t len = br.peek(16);
        const uint32_t nlen = (uint32_t)(br.buf >> 16) & 0xffffu;
        if (len != ((~nlen) & 0xffffu)) return fail(ST_FMT_LEN_NLEN, len, nlen);
        br.drop(32);
        const uint64_t p = stream_bit_pos() >> 3;  // byte offset of the raw data in the stream
        if (p + len > in_len) return fail(ST_TRUNCATED, 0, 0);
        // Monad.hs:265-293 nextBlock + Monad.hs:317-322 emitBlock: raw bytes straight from HBM to the ring
        const uint32_t lane = lane_id();
        uint32_t done = 0;
        constexpr uint32_t PIECE = RING / 4u;
        while (done < len) {
            uint32_t piece = len - done < PIECE ? len - done : PIECE;
            if (op + piece - flushed > RING) flush_to(op & ~(uint64_t)15u);
            for (uint32_t k = lane; k < piece; k += PZG_WAVE)
                L.ring[((uint32_t)op + k) & RMASK] = in[p + done + k];
            op += piece;
            done += piece;
        }
        maybe_flush();
        in_byte0 = p + len;
        br.start(in, in_len, in_byte0);
        return ST_OK;
    }

    // ---- Deflate.hs:79-82,241-251: the fixed code ---------------------------------------------------
    PZG_FN void load_fixed_tables()
    {
        if (fixed_loaded) return;
        const uint32_t lane = lane_id();
        for (uint32_t s = lane; s < 288u; s += PZG_WAVE)
            L.lens[s] = (uint8_t)(s <= 143u ? 8u : s <= 255u ? 9u : s <= 279u ? 7u : 8u);
        for (uint32_t s = lane; s < 32u; s += PZG_WAVE) L.lens[288u + s] = 5u;
        build_table<LIT_BITS, TREE_LITLEN>(L.lens, 288u, L.lit_lut, L.lit_sorted, &L.lit_meta, &lit_e15);
        build_table<DIST_BITS, TREE_DIST>(L.lens + 288u, 32u, L.dist_lut, L.dist_sorted, &L.dist_meta, &dist_e15);
        fixed_loaded = 1;
    }

    // ---- Deflate.hs:83-101,124-156: dynamic block header ---------------------------------------------
    PZG_FN int dynamic_header(uint32_t block_bit)
    {
        const uint32_t lane = lane_id();
        br.refill();
        if (br.avail() < 14) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t hlit = 257u + br.peek(5);
        const uint32_t hdist = 1u + ((uint32_t)(br.buf >> 5) & 31u);
        const uint32_t hclen = 4u + ((uint32_t)(br.buf >> 10) & 15u);
        br.drop(14);
        // hclen x 3-bit lengths in codeLengthOrder (Deflate.hs:87-88,290-292)
        for (uint32_t i = lane; i < 20u; i += PZG_WAVE) L.cl_lens[i] = 0;
        wave_sync();
        for (uint32_t i = 0; i < hclen; ++i) {
            br.refill();
            if (br.avail() < 3) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t v = br.peek(3);
            br.drop(3);
            // codeLengthOrder = 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 packed 5 bits each
            const uint64_t ORD_LO = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 |
                                    9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
            const uint64_t ORD_HI = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
            const uint32_t sym = i < 12u ? (uint32_t)(ORD_LO >> (5u * i)) & 31u : (uint32_t)(ORD_HI >> (5u * (i - 12u))) & 31u;
            if (lane == 0u) L.cl_lens[sym] = (uint8_t)v;
        }
        uint32_t cl_e15;
        if (!build_table<CL_BITS, TREE_CODELEN>(L.cl_lens, 19u, L.dist_lut, nullptr, nullptr, &cl_e15))
            return fail(ST_HUFF_BUILD, TREE_CODELEN, block_bit);
        // getCodeLengths (Deflate.hs:124-156) over HLIT+HDIST as ONE sequence
        const uint32_t maxl = hlit + hdist;
        uint32_t n = 0, prev = 0;
        while (n < maxl) {
            br.refill();
            const uint32_t e = uni(L.dist_lut[br.peek(CL_BITS)]);
            if (int st = check_entry(e, 0)) return st;
            const uint32_t sym = ent_val(e), cn = ent_n(e), ce = ent_e(e);
            if (br.avail() < (int64_t)(cn + ce)) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t extra = ((uint32_t)(br.buf >> cn)) & ((1u << ce) - 1u);
            br.drop(cn + ce);
            if (sym <= 15u) {
                if (lane == 0u) L.lens[n] = (uint8_t)sym;
                n++;
                prev = sym;
                continue;
            }
            uint32_t num, val;
            if (sym == 16u) {  // repeat previous 3-6 times; with no predecessor prev is 0 (Deflate.hs:91,136-139)
                num = 3u + extra;
                val = prev;
            } else if (sym == 17u) {
                num = 3u + extra;
                val = 0;
                prev = 0;
            } else {
                num = 11u + extra;
                val = 0;
                prev = 0;
            }
            // repeats that run past HLIT+HDIST are accepted and spill into extra distance symbols (Deflate.hs:132,96-97)
            for (uint32_t k = lane; k < num; k += PZG_WAVE) L.lens[n + k] = (uint8_t)val;
            n += num;
        }
        wave_sync();
        // litTree first, then distTree (Deflate.hs:98-99): errors surface in that order
        if (!build_table<LIT_BITS, TREE_LITLEN>(L.lens, hlit, L.lit_lut, L.lit_sorted, &L.lit_meta, &lit_e15))
            return fail(ST_HUFF_BUILD, TREE_LITLEN, block_bit);
        if (!build_table<DIST_BITS, TREE_DIST>(L.lens + hlit, n - hlit, L.dist_lut, L.dist_sorted, &L.dist_meta, &dist_e15))
            return fail(ST_HUFF_BUILD, TREE_DIST, block_bit);
        fixed_loaded = 0;
        return ST_OK;
    }

    // ---- Zlib.hs:53-69 inflateWithHeaders + Deflate.hs:39-63 inflate ---------------------------------
    PZG_FN void run(const uint8_t *in_, uint64_t in_len_, uint8_t *out_, uint64_t cap_, StreamResult *res)
    {
        in = in_;
        in_len = in_len_;
        out = out_;
        cap = cap_;
        op = 0;
        flushed = 0;
        adler_a = 1;
        adler_b = 0;
        lit_e15 = dist_e15 = 0;
        fixed_loaded = 0;
        status = ST_OK;
        detail0 = detail1 = 0;
        in_byte0 = 0;
        br.start(in, in_len, 0);
        decode();
        uint64_t used_bits = stream_bit_pos();
        uint64_t used = (used_bits + 7u) >> 3;
        if (used > in_len) used = in_len;
        if (status == ST_OK && op > cap) status = ST_OUT_TOO_SMALL;
        res->status = status;
        res->detail0 = detail0;
        res->detail1 = detail1;
        res->adler = (adler_b << 16) | adler_a;
        res->out_len = op;
        res->in_used = used;
    }

    PZG_FN int decode()
    {
        // Zlib.hs:55-67: CMF, FLG; FCHECK, then CM, then CINFO
        if (br.avail() < 16) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t cmf = br.peek(8);
        const uint32_t flg = (uint32_t)(br.buf >> 8) & 0xffu;
        br.drop(16);
        if (((cmf << 8) | flg) % 31u != 0u) return fail(ST_HDR_FCHECK, (cmf << 8) | flg, 0);
        if ((cmf & 15u) != 8u) return fail(ST_HDR_METHOD, cmf & 15u, 0);
        if ((cmf >> 4) > 7u) return fail(ST_HDR_WINDOW, cmf >> 4, 0);
        if (flg & 0x20u) {  // Zlib.hs:68: skip DICTID, carry on with an empty history
            br.refill();
            if (br.avail() < 32) return fail(ST_TRUNCATED, 0, 0);
            br.drop(32);
        }
        for (;;) {  // Deflate.hs:45-50 go
            br.refill();
            const uint32_t block_bit = (uint32_t)stream_bit_pos();
            if (br.avail() < 3) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t bfinal = br.peek(1);
            const uint32_t btype = (uint32_t)(br.buf >> 1) & 3u;
            br.drop(3);
            int st;
            if (btype == 0u) {
                st = stored_block();
            } else if (btype == 1u) {
                load_fixed_tables();
                st = token_loop();
            } else if (btype == 2u) {
                st = dynamic_header(block_bit);
                if (st == ST_OK) st = token_loop();
            } else {
                st = fail(ST_FMT_BTYPE, 3, 0);
            }
            if (st != ST_OK) return st;
            if (bfinal) break;
        }
        // Deflate.hs:52-63 checkChecksum: align, fold the rest of the window, compare big-endian
        flush_to(op);
        br.drop(br.cnt & 7u);
        br.refill();
        if (br.avail() < 32) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t t = (uint32_t)br.buf;
        const uint32_t theirs = (t << 24) | ((t & 0xff00u) << 8) | ((t >> 8) & 0xff00u) | (t >> 24);
        br.drop(32);
        const uint32_t ours = (adler_b << 16) | adler_a;
        if (theirs != ours) return fail(ST_CHECKSUM, theirs, ours);
        return ST_OK;
    }
};

}  // namespace pzg
