// inflate_core.h -- one zlib stream decoded by one wavefront.
//
// MI355X counterpart of pure-zlib's whole decode stack:
//   Zlib.hs:53-69      inflateWithHeaders   -> Decoder::decode() prologue
//   Deflate.hs:39-63   inflate/checkChecksum-> Decoder::decode() block loop + trailer
//   Deflate.hs:65-104  inflateBlock         -> stored_block() / dynamic_header() / load_fixed_tables()
//   Deflate.hs:106-120 runInflate           -> token_loop(): hot_loop() [window2_decode() + walk_half() -> token queue ->
//                                             emit_body<FAST>()], the general window_append2() / emit_segment(), token_step_checked()
//   Deflate.hs:124-156 getCodeLengths       -> dynamic_header()
//   Deflate.hs:160-237 length/distance arrays -> litlen_entry()/dist_entry() (closed forms)
//   Deflate.hs:255-292 computeCodeValues    -> build_table() (canonical codes, wave-parallel)
//   HuffmanTree.hs     binary trie          -> multi-level LDS table: a direct 2^P LUT indexed by the next P stream
//                                             bits, second-level tables for codes longer than P, and behind
//                                             them an exact canonical first-code walk + ballot scan of the lengths
//   Monad.hs:203-307   bit/byte reader      -> BitReader: coalesced dword chunks held one dword per
//                                             lane, a per-wave bit cursor, v_readlane to fetch
//   OutputWindow.hs    128 KiB flat window  -> 2^RING_BITS LDS ring (+ far reads of the stream's own flushed output when
//                                             RING_BITS < 15), lane-cooperative LZ77 copy
//   Adler32.hs         per-byte checksum    -> folded into the ring->HBM flush as a wave reduction
//
// The hot loop is wave-parallel: lane k speculatively decodes the complete tokens (literal, or
// length + distance with their extra bits) that would start k and k + 64 bits ahead of the cursor --
// two LDS lookups each, for all 128 offsets at once -- and a scalar walk then follows the real chain
// from offset 0 with v_readlane, so no LDS round trip is paid per token.  The real tokens are
// compacted onto a per-wave token queue; emit_body() turns the queue's head into <= 128 output
// bytes in two passes of one ring gather and one ring store each.  hot_loop() runs these for as long as nothing
// needs code with a lane-dependent branch: it is written (and the kernels are built) so that the compiler's
// structurizer leaves its control flow alone -- see the comment there; that alone is worth 20 % of the kernel's speed.
//
// The same source compiles as a host program for the CPU model tests (see wave.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "wave.h"

// Diagnostic build only (-DPZG_PROFILE): per-phase cycle stamps; never compiled into libpzg.so.
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS && defined(PZG_PROFILE_HDR)  // slots 8-11 = the phases of dynamic_header() instead of emit's
#define PZG_T0(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define PZG_ACC(slot, var) ((slot) >= 8 && (slot) <= 11 ? (void)var : (void)(prof[slot] += __builtin_amdgcn_s_memtime() - var))
#define PZG_ACCW(slot, var) ((void)var)
#define PZG_HACC(slot, var) prof[slot] += __builtin_amdgcn_s_memtime() - var
#elif defined(PZG_PROFILE) && PZG_DEVICE_PASS && defined(PZG_PROFILE_HOT)  // slots 8-10 = inside hot_loop(): windows, segments; the rare window path
#define PZG_T0(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define PZG_ACC(slot, var) ((slot) >= 8 && (slot) <= 11 ? (void)var : (void)(prof[slot] += __builtin_amdgcn_s_memtime() - var))
#define PZG_ACCW(slot, var) ((void)var)
#define PZG_HACC(slot, var)
#define PZG_HOT_ACC(slot, var) prof[slot] += __builtin_amdgcn_s_memtime() - var
#elif defined(PZG_PROFILE) && PZG_DEVICE_PASS
#define PZG_T0(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#define PZG_ACC(slot, var) prof[slot] += __builtin_amdgcn_s_memtime() - var
#define PZG_ACCW(slot, var) (__builtin_amdgcn_s_waitcnt(0xc07f), prof[slot] += __builtin_amdgcn_s_memtime() - var)  // after lgkmcnt(0)
#define PZG_HACC(slot, var)
#else
#define PZG_HACC(slot, var)
#define PZG_T0(var)
#define PZG_ACC(slot, var)
#define PZG_ACCW(slot, var)
#endif
// Diagnostic build only (-DPZG_MARKS): named comments in the device assembly (tests/tools/asm_regions.py counts between them)
#if defined(PZG_MARKS) && PZG_DEVICE_PASS
#define PZG_MARK(name) asm volatile("; ##MARK " name)
#else
#define PZG_MARK(name)
#endif
#ifndef PZG_HOT_ACC
#define PZG_HOT_ACC(slot, var)
#endif
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS && defined(PZG_PROFILE_HOT)  // the parts of a group (seq_group), every wait charged where it stands
#define PZG_SEQ_T0(var) uint64_t var = __builtin_amdgcn_s_memtime()
#define PZG_SEQ_ACC(slot, var) (__builtin_amdgcn_s_waitcnt(0x0070), prof[slot] += __builtin_amdgcn_s_memtime() - var, var = __builtin_amdgcn_s_memtime())
#else
#define PZG_SEQ_T0(var)
#define PZG_SEQ_ACC(slot, var)
#endif

// Lab build of the HOST model only (-DPZG_STATS, tests/tools/model_stats.py): event counts of the token loop
#if defined(PZG_STATS) && !PZG_DEVICE_PASS
struct PzgStats { unsigned long long v[32]; };
extern PzgStats pzg_stats;
#define PZG_STAT(i, n) (pzg_stats.v[i] += (unsigned long long)(n))
#else
#define PZG_STAT(i, n) ((void)0)
#endif
#ifndef PZG_RES_HOT_LOOP
#define PZG_RES_HOT_LOOP 1
#endif
#ifndef PZG_WALK_UNROLL
#define PZG_WALK_UNROLL 8
#endif
#ifndef PZG_HOT_ALIGN
#define PZG_HOT_ALIGN 6
#endif
#ifndef PZG_WALK_EXIT_ALIGN
#define PZG_WALK_EXIT_ALIGN 2
#endif
#define PZG_STR2(x) #x
#define PZG_STR(x) PZG_STR2(x)

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
    ST_OUT_TOO_SMALL = 14,
    ST_GZIP_HEADER = 18,  // extension (RFC 1952): detail0 = 1 magic, 2 method, 3 reserved flag bits, 4 header CRC16
    ST_GZIP_ISIZE = 19,   // extension: detail0 = ISIZE in the trailer, detail1 = bytes produced mod 2^32
    ST_DICT = 20,         // extension (PZG_FDICT): detail0 = DICTID of the stream, detail1 = Adler-32 of the dictionary supplied
    ST_NEED_INPUT = 101,  // resumable decoder: every complete token of the input so far has been decoded; more input is needed
    ST_OUT_FULL = 102,    // resumable decoder: the output room of this call is used up; call again with what is left of the input
    ST_RETRY_FULL_RING = 100,  // internal: a small-ring launch met an output larger than its capacity; the 32 KiB ring kernel redoes the stream
    ST_BUNDLE_TODO = 103      // internal (bundles, bundle_core.h): the stream is the ordinary kernel's
};

enum { TREE_CODELEN = 0, TREE_LITLEN = 1, TREE_DIST = 2 };

#ifndef PZG_LIT_BITS
#define PZG_LIT_BITS 8
#endif
constexpr int LIT_BITS = PZG_LIT_BITS;  // primary literal/length LUT: 2^8 x 4 B = 1 KiB (LDS is what bounds residency)
#ifndef PZG_SUB_ENTRIES
#define PZG_SUB_ENTRIES 252
#endif
// pool of second-level entries for literal/length (then distance) codes longer than the primary tables resolve: 252 fills the
// wave's LDS up to the 6 KiB that 26 waves per CU leave each (188 -> 252: literal-heavy data +12 %); an entry's index is 8 bits
constexpr uint32_t SUB_ENTRIES = PZG_SUB_ENTRIES;
static_assert(SUB_ENTRIES <= 256u, "K_SUB entries hold an 8-bit pool index");
#ifndef PZG_SUB_BITS_MAX
#define PZG_SUB_BITS_MAX 5
#endif
constexpr uint32_t SUB_BITS_MAX = PZG_SUB_BITS_MAX;    // a second-level table resolves at most this many further bits
#ifndef PZG_SUB_MIN
#define PZG_SUB_MIN 3
#endif
constexpr uint32_t SUB_MIN_PREFIXES = PZG_SUB_MIN;  // fewer long prefixes than this: their tokens are too rare to pay for a second lookup in the windows
constexpr int DIST_BITS = 8;  // primary distance LUT:        2^8  x 4 B = 1 KiB
constexpr int CL_BITS = 7;    // code-length code: max length 7, the LUT is exhaustive
constexpr uint32_t ADLER_MOD = 65521u;

constexpr int MAX_LIT_SYMS = 288;        // HLIT <= 288
constexpr int MAX_DIST_SYMS = 32 + 138;  // HDIST <= 32 plus a code-length repeat overrun (Deflate.hs:132)
constexpr int MAX_LENS = 288 + 32 + 138;

// LUT entries (literal/length, distance, second-level and code-length tables), 32 bits.  The layout is what the
// windows' speculative decode (spec_finish) needs in the fewest vector instructions: a shift or bit-field
// operand is the low five bits of a register, so every field that is used as one starts a byte.
//   token entries (bit 7 clear)
//     literal         [4:0] n              [15:8] byte   [24:16] 1                   -- the queued token itself
//     length base     [4:0] n + extra bits [12:8] n      [24:16] base length  [31] 1
//     distance base   [4:0] n + extra bits [12:8] n      [31:16] base distance
//     code-length sym [4:0] n              [12:8] extra bits          [31:16] symbol (dynamic_header only)
//   stoppers (bit 7 set): a window's walk ends there and token_step_checked() takes over
//     [4:0] n  [7] 1  [11:8] kind  [29:16] value  [30] 1 for K_SUB only (the only entries >= 2^30 as signed numbers)
enum : uint32_t {
    K_LIT = 0,           // (token entry, bit 31 clear) literal byte or code-length symbol
    K_BASE = 1,          // (token entry) base length (bit 31 set) / base distance, extra bits follow
    K_EOB = 2,           // symbol 256
    K_LONG = 3,          // code longer than the tables resolve: decode_long()
    K_EMPTY_BRANCH = 4,  // n = depth at which the reference's walk reaches HuffmanEmpty
    K_EMPTY_TREE = 5,    // the tree has no codes at all
    K_BADSYM = 6,        // symbol 286/287 or distance symbol >= 30: value = symbol
    K_SUB = 7            // second-level table at sub[value], indexed by the next n bits
};
constexpr uint32_t ENT_STOP = 0x80u;  // bit 7: the byte sum tb of spec_finish() is then >= 128, which the walk tests for
constexpr uint32_t ENT_MATCH = 0x80000000u;
constexpr uint32_t ENT_SUB = 0x40000000u;  // K_SUB entries: found by ONE signed compare in the windows' decode

PZG_FN uint32_t mk_stop(uint32_t n, uint32_t kind, uint32_t value) { return n | ENT_STOP | (kind << 8) | (value << 16); }
PZG_FN bool ent_is_stop(uint32_t x) { return (x & ENT_STOP) != 0u; }
PZG_FN uint32_t ent_stop_kind(uint32_t x) { return (x >> 8) & 15u; }   // stoppers only
PZG_FN uint32_t ent_n(uint32_t x) { return x & 31u; }                   // stoppers, literals, code-length symbols
PZG_FN uint32_t ent_val(uint32_t x) { return x >> 16; }                 // stoppers but K_SUB (value), distance base, code-length symbol
PZG_FN uint32_t ent_sub_index(uint32_t x) { return (x >> 16) & 0xffu; } // K_SUB: first entry of the second-level table in the pool
PZG_FN uint32_t ent_base_n(uint32_t x) { return (x >> 8) & 31u; }       // length / distance base: code bits
PZG_FN uint32_t ent_base_tot(uint32_t x) { return x & 31u; }            // ... code bits + extra bits
PZG_FN uint32_t ent_len_base(uint32_t x) { return (x >> 16) & 511u; }   // length base
PZG_FN uint32_t ent_lit_byte(uint32_t x) { return (x >> 8) & 255u; }
PZG_FN uint32_t ent_kind_lit(uint32_t x) { return ent_is_stop(x) ? ent_stop_kind(x) : (x & ENT_MATCH) ? (uint32_t)K_BASE : (uint32_t)K_LIT; }
PZG_FN uint32_t ent_kind_dist(uint32_t x) { return ent_is_stop(x) ? ent_stop_kind(x) : (uint32_t)K_BASE; }

// Deflate.hs:164-196 lengthArray as a closed form: symbol 257..285 -> (base, extra)
PZG_FN uint32_t litlen_entry(uint32_t sym, uint32_t n)
{
    if (sym < 256u) return n | (sym << 8) | (1u << 16);
    if (sym == 256u) return mk_stop(n, K_EOB, 0);
    if (sym > 285u) return mk_stop(n, K_BADSYM, sym);
    uint32_t i = sym - 257u, e = 0, base;
    if (i < 8u) base = 3u + i;
    else if (i == 28u) base = 258u;
    else {
        e = (i >> 2) - 1u;
        base = 3u + ((4u + (i & 3u)) << e);
    }
    return (n + e) | (n << 8) | (base << 16) | ENT_MATCH;
}

// Deflate.hs:203-237 distanceArray as a closed form: code 0..29 -> (base, extra)
PZG_FN uint32_t dist_entry(uint32_t sym, uint32_t n)
{
    if (sym > 29u) return mk_stop(n, K_BADSYM, sym);
    uint32_t e = 0, base = 1u + sym;
    if (sym >= 4u) {
        e = (sym >> 1) - 1u;
        base = 1u + ((2u + (sym & 1u)) << e);
    }
    return (n + e) | (n << 8) | (base << 16);
}

// code-length alphabet (Deflate.hs:131-149): 0..15 literal lengths, 16/17/18 repeats with 2/3/7 extra bits
PZG_FN uint32_t codelen_entry(uint32_t sym, uint32_t n)
{
    uint32_t e = sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
    return n | (e << 8) | (sym << 16);
}
PZG_FN uint32_t ent_cl_extra(uint32_t x) { return (x >> 8) & 15u; }

// ---- LDS image of one wave ------------------------------------------------------------------
struct TreeMeta {          // second-level (canonical) decode tables, index = code length 1..15
    uint16_t count[16];    // symbols of that length
    uint16_t first[16];    // first canonical code of that length
};

#ifndef PZG_DMA_PREFETCH
#define PZG_DMA_PREFETCH 1
#endif

template <int RING_BITS>
struct alignas(16) WaveLds {
#if PZG_DMA_PREFETCH
    uint32_t pf[64];                     // input prefetch, written by global_load_lds (must stay the first member: LDS offset 0)
#endif
    uint8_t ring[1u << RING_BITS];       // OutputWindow: the last 2^RING_BITS bytes produced
    uint32_t lit_lut[1u << LIT_BITS];    // HuffmanTree (literal/length), level 1
    uint32_t dist_lut[1u << DIST_BITS];  // HuffmanTree (distance), level 1; the code-length LUT while a header is read
    uint32_t sub[SUB_ENTRIES];           // HuffmanTree (literal/length), level 2 (see build_table)
    TreeMeta lit_meta;
    TreeMeta dist_meta;
    uint32_t cnt[16];                    // histogram / running-rank scratch for build_table
    uint8_t lens[MAX_LENS + 22];         // code lengths of the block being set up
    uint8_t cl_lens[20];                 // code-length code lengths in symbol order
    uint32_t fixed_ready;                // 0x51DF1BED while lit/dist tables hold the fixed code: survives from one stream to the
                                         // next on a persistent wave, so a batch of fixed-Huffman streams builds it once per wave
    uint8_t dump[64 + 12];               // where masked-off lanes store (see sel_store): keeps hot loops free of lane-dependent branches
    uint32_t strip_back;                 // the strips' run-up in bits, as the wave's last spans taught it (strip_span): survives from one
                                         // stream to the next on a persistent wave, like fixed_ready; anything out of range means "the default"
};

// ---- result of one stream -------------------------------------------------------------------
struct StreamResult {
    int32_t status;
    uint32_t detail0, detail1;
    uint32_t adler;
    uint32_t gz_crc;   // gzip only: the CRC-32 the output must have according to the members' trailers
    uint64_t out_len;
    uint64_t in_used;
};

// ---- resumable decoding (decompressIncremental, Monad.hs:163-197): what a suspended decoder keeps in HBM -------
// The scalars below, the token queue, and behind them the wave's whole LDS image (tables + the 32 KiB ring: the
// resumable kernel is the RING_BITS = 15 instance, so the window needs nothing but LDS).
enum : uint32_t { PH_HEADER = 0, PH_BLOCK = 1, PH_STORED = 2, PH_TOKENS = 3, PH_TRAILER = 4, PH_DONE = 5 };
struct ResumeState {
    uint32_t phase;        // where decoding resumes (a fresh decoder is all zeros)
    uint32_t bfinal;       // the block being decoded is the last one
    uint32_t stored_left;  // PH_STORED: bytes of the stored block still to copy
    uint32_t deferred;     // PH_TOKENS: 0, or the outcome (end of block / an error status) met while the queue was not yet drained
    uint32_t qn;           // tokens waiting in QT
    uint32_t bit_skip;     // bits of the first byte of the next input that are already consumed
    uint32_t ow, chunks;   // the reference's window fill (OutputWindow.hs owNext) and the 32 KiB chunks it has published so far
    uint64_t op;           // bytes produced so far
    uint32_t adler_a, adler_b, lit_e15, dist_e15, lit_n, dist_n, use_sub, lit_sub_used;
    int32_t status;        // a terminal status once the decoder has failed (PH_DONE)
    uint32_t detail0, detail1;
    uint32_t dist_sub_used;  // second-level entries of the distance code (strip_span() asks: ADVICE r5)
    uint64_t in_total;     // input bytes consumed by the earlier calls (positions in error details count from the stream start)
    uint32_t QT[64];
};

// A suspended decoder's slot in HBM: ResumeState | the wave's LDS image | (small rings) its 32 KiB history.
template <int RING_BITS>
struct ResumeSlot {
    static constexpr size_t IMAGE_OFF = sizeof(ResumeState);
    static constexpr size_t HIST_OFF = (sizeof(ResumeState) + sizeof(WaveLds<RING_BITS>) + 255u) & ~(size_t)255u;
    static constexpr size_t BYTES = HIST_OFF + (RING_BITS < 15 ? 32768u : 0u);
};

// CRC-32 (reflected, poly 0xedb88320) as a polynomial over GF(2): a * b mod P, and the CRC of "A followed by n more
// bytes that are B" from the two finalized CRCs (zlib's crc32_combine): shift A by 8n bits, add B.
PZG_FN uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma nounroll
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (0xedb88320u & (0u - (b & 1u)));
    }
    return p;
}
PZG_FN uint32_t crc32_append(uint32_t crc_a, uint32_t crc_b, uint64_t len_b)
{
    uint32_t pw = 0x80000000u, sq = 0x00800000u;  // x^0, x^8
#pragma nounroll
    for (uint64_t e = len_b; e != 0; e >>= 1) {
        if (e & 1u) pw = gf2_mul(pw, sq);
        sq = gf2_mul(sq, sq);
    }
    return gf2_mul(crc_a, pw) ^ crc_b;
}

// ---- Monad.hs:203-307: the bit reader ---------------------------------------------------------
// The compressed stream is read as aligned dwords, 64 at a time: lane l of `cur` holds dword
// chunk0+l (one fully coalesced 256-byte load per chunk), `nxt` the following chunk, prefetched.
// `pos` is the per-wave bit cursor; dword(i) fetches a wave-uniform dword with v_readlane.
// Bits are consumed LSB-first (Monad.hs:224-230).
struct BitReader {
    // Round 4: consecutive chunks OVERLAP -- `cur` holds 64 dwords but the cursor moves on to the next chunk after
    // STRIDE = 56 of them, so the 5 dwords a 128-bit window's last lane needs behind the cursor's own (dword 57 + 4 at most)
    // are always in `cur`: the windows never gather from `nxt` (two scalar instructions per window and a rare path less,
    // for 14 % more input loads, which the L2 serves).
    static constexpr uint32_t STRIDE = 56u, STRIDE_BITS = 32u * STRIDE;
    const uint32_t *base;  // 4-byte aligned address at or below the stream start
    uint32_t ndw;          // dwords covering [base, stream end)
    uint32_t mis_bits;     // 8 * (stream start - base)
    uint64_t end_rel;      // mis_bits + 8 * stream length: first bit (relative to base) past the stream
    uint32_t win_end;      // a cursor in a dword below this index has >= 192 stream bits in front of it
    // The cursor is kept as (chunk0, rp): the hot loops only ever touch the 32-bit rp.
    uint32_t chunk0;       // dword index of the 64-dword chunk the cursor is in (a multiple of STRIDE; lane 0 of `cur`)
    uint32_t rp;           // next unread bit, relative to bit 32 * chunk0; slide() keeps it below STRIDE_BITS
    int32_t rp_ok1, rp_ok2;  // rp below these (signed): a 64-bit / a 128-bit window may run (set_limits)
#if PZG_DEVICE_PASS
    uint32_t cur, nxt;     // per-lane: dwords chunk0 + lane and chunk0 + STRIDE + lane
#if PZG_DMA_PREFETCH
    // Two chunks ahead: the chunk after `nxt` is fetched straight into LDS (global_load_lds: no VGPR, so no
    // register copy can force a wait for it) and picked up one slide later.
    const volatile uint32_t *pf;  // WaveLds::pf (LDS offset 0)
    PZG_FN void dma_prefetch(uint32_t c0) const
    {
        if (c0 >= ndw) return;  // wave-uniform
        const uint32_t i = c0 + lane_id();
        const uint32_t *p = base + (i < ndw ? i : ndw - 1u);
        asm volatile("s_mov_b32 m0, 0\n\tglobal_load_lds_dword %0, off" ::"v"(p) : "memory");  // (m0 is reserved by the compiler, never live across statements here)
    }
    PZG_FN uint32_t take_prefetch() const
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t l = lane_id();
        asm volatile("" : "+v"(l));  // (opaque: or 4 * lane, as a 64-bit offset, is computed in the kernel's prologue and kept -- in the gzip instance: spilled -- for its whole life)
        const uint32_t v = pf[l];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the buffer is free again before the next fetch is issued
        return v;
    }
#endif
#endif

    PZG_FN uint64_t pos() const { return ((uint64_t)chunk0 << 5) + rp; }  // next unread bit, relative to base

    PZG_FN uint32_t load_dw(uint32_t i) const { return i < ndw ? base[i] : 0u; }

    // lane l's dword of the 64-dword chunk starting at c0 (wave-uniform), zero past the stream.
    // Branch-free per lane (clamped index + select): a lane-dependent branch here would sit inside
    // the loops that carry the wave-uniform decoder state and make the compiler treat that state as
    // divergent (VALU + exec-mask loops instead of SALU + s_cbranch).
    PZG_FN uint32_t load_chunk(uint32_t c0) const { return zero_past_end(load_chunk_raw(c0), c0); }
    // The load and the masking are separate so that `nxt` can stay an outstanding load (a prefetch):
    // nothing waits for it until slide() promotes it to `cur`.
    PZG_FN uint32_t load_chunk_raw(uint32_t c0) const
    {
        if (c0 >= ndw) return 0u;  // wave-uniform
        const uint32_t i = c0 + lane_id();
        return base[i < ndw ? i : ndw - 1u];
    }
    PZG_FN uint32_t zero_past_end(uint32_t v, uint32_t c0) const { return c0 + lane_id() < ndw ? v : 0u; }

    // the window tests, relative to the current chunk (recomputed once per 2048 bits, tested once per window):
    //   64-bit window   dword index of the cursor     < win_end
    //   128-bit window  dword index of the cursor + 3 < win_end   (10 whole dwords follow)
    PZG_FN void set_limits()
    {
        // (32-bit arithmetic only: a 64-bit compare would be a vector instruction and leave the limits in vector registers)
        uint32_t d = win_end >= chunk0 ? win_end - chunk0 : 0u;
        d = d > 4096u ? 4096u : d;  // (rp stays below 2048 + one window: anything past that is "yes")
        rp_ok1 = (int32_t)uni(d * 32u);  // (pinned to scalar registers: the hot loop compares against them once per window)
        rp_ok2 = (int32_t)uni(d * 32u - 96u);
    }

    PZG_FN void start(const uint8_t *in, uint64_t in_len, uint64_t byte_pos)
    {
        const uint8_t *p = in + byte_pos;
        uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
        uint64_t remain = in_len - byte_pos;
        base = (const uint32_t *)(p - mis);
        mis_bits = mis * 8u;
        ndw = (uint32_t)((mis + remain + 3u) >> 2);
        end_rel = (uint64_t)mis_bits + remain * 8u;
        win_end = (end_rel >> 5) >= 6u ? (uint32_t)(end_rel >> 5) - 6u : 0u;
        chunk0 = 0;
        rp = mis_bits;
        set_limits();
#if PZG_DEVICE_PASS
        // (the chunk indices pass through an opaque statement: as literal constants, lane + 64 and lane + 128 are computed once
        // in the kernel's prologue and held in two vector registers for its whole life -- or spilled, in the gzip instance)
        uint32_t c1 = STRIDE, c2 = 2u * STRIDE;
        asm volatile("" : "+s"(c1), "+s"(c2));
        cur = load_chunk(0u);
        nxt = load_chunk_raw(c1);  // masked when it becomes `cur`
#if PZG_DMA_PREFETCH
        dma_prefetch(c2);
#endif
#endif
    }

    // wave-uniform dword i; on the device i must lie in [chunk0, chunk0 + STRIDE + 64)
    PZG_FN uint32_t dword(uint32_t i) const
    {
#if PZG_DEVICE_PASS
        const uint32_t a = read_lane(cur, (i - chunk0) & 63u), b = read_lane(nxt, (i - chunk0 - STRIDE) & 63u);
        return (i - chunk0) < 64u ? a : i < ndw ? b : 0u;
#else
        return load_dw(i);
#endif
    }

    // one chunk forward
    PZG_FN void step_chunk()
    {
        rp -= STRIDE_BITS;
        chunk0 += STRIDE;
#if PZG_DEVICE_PASS
        cur = zero_past_end(nxt, chunk0);
#if PZG_DMA_PREFETCH
        nxt = take_prefetch();
        dma_prefetch(chunk0 + 2u * STRIDE);
#else
        nxt = load_chunk_raw(chunk0 + STRIDE);
#endif
#endif
        set_limits();
    }
    // keep the cursor's dword inside `cur`
    PZG_FN void slide()
    {
        while (rp >= STRIDE_BITS) step_chunk();
    }

    PZG_FN int64_t avail() const { return (int64_t)end_rel - (int64_t)pos(); }
    // cheap sufficient test for avail() >= 192 (7 whole dwords follow the cursor's dword index)
    PZG_FN bool window_ok() const { return (int32_t)rp < (int32_t)uni((uint32_t)rp_ok1); }
    // same for a 128-bit window (10 whole dwords follow)
    PZG_FN bool window2_ok() const { return (int32_t)rp < (int32_t)uni((uint32_t)rp_ok2); }  // (uni: keeps the compare scalar)

    // the next 32 bits (bits past the stream end read as whatever follows; callers check avail())
    PZG_FN uint32_t peek32() const
    {
        const uint32_t i = chunk0 + (rp >> 5);
        return funnel_uniform(dword(i + 1u), dword(i), rp & 31u);
    }
    PZG_FN void drop(uint32_t n)
    {
        rp += n;
        slide();
    }
    // drop() for n < STRIDE_BITS from a cursor slide() has already placed: at most one chunk step, written
    // without a loop so the prefetch into `nxt` stays an outstanding load (no copy of it is needed).
    PZG_FN void drop_short(uint32_t n)
    {
        rp += n;
        if (__builtin_expect(rp >= STRIDE_BITS, 0)) step_chunk();
    }
    PZG_FN void align_to_byte() { drop((8u - (rp & 7u)) & 7u); }  // (32 * chunk0 is a multiple of 8)
};

// ---- decoder state (all wave-uniform) -----------------------------------------------------------
// GZIP: the streams are RFC 1952 members (an extension; its own kernel instances, so the zlib ones carry none of it)
// RES: the resumable instance (decompressIncremental): suspends when the input or the output room runs out
template <int RING_BITS, bool GZIP = false, bool RES = false>
struct Decoder {
    static constexpr uint32_t RING = 1u << RING_BITS;
    static constexpr uint32_t RMASK = RING - 1u;
    static constexpr uint32_t FLUSH_AT = RING - 1024u;
    static constexpr int WINDOW_MIN_BITS = 192;  // 5 dwords + slack: every token of a window is real data

    WaveLds<RING_BITS> &L;
    const uint8_t *in;
    uint64_t in_len;
    uint8_t *out;
    uint64_t cap;
    BitReader br;
    uint64_t in_byte0;  // byte offset in the stream at which br was (re)started
    uint64_t op;        // bytes produced
    uint64_t flushed;   // bytes already written to HBM and folded into the Adler state
    uint32_t adler_a, adler_b;
    uint32_t lit_e15, dist_e15;  // Kraft totals in 2^-15 units (0 = empty tree)
    uint32_t lit_n, dist_n;      // symbols of the current block's two codes: lens[0..lit_n) and lens[lit_n..lit_n+dist_n)
    uint32_t use_sub;            // the block's literal/length code has enough long prefixes: windows do the second lookup
    uint32_t lit_sub_used;       // second-level entries taken by the literal/length code (the distance code's follow)
    uint32_t dist_sub_used;      // ... by the distance code
    // strips (strip_span): this wave's scratch in HBM (null: no strips), the sequence records of the span being emitted
    uint32_t *strip;
    uint32_t s_rd;                  // the group's first record is record number s_rd of the span
    uint32_t s_total, s_cnt;        // records of the span; lanes whose regions hold any
    LaneVec<uint32_t> CRIDX, CSTA;  // the t-th region that holds records: its lane, and the span's records in front of it (seq_index)
    LaneVec<uint32_t> QTN;          // the next group's records (seq_refill): lane j holds record s_rd + j
    LaneVec<uint32_t> QINFO;        // ... [7:0] the region it lies in, [15:8] the first lane of this group that lies in the same region,
                                    // [16] that region is the one the group starts in
    uint32_t s_qn;                  // ... how many of them there are (<= 64)
    uint32_t s_lc;                  // literal bytes of the region record s_rd lies in that the records in front of s_rd account for
    int32_t status;
    uint32_t detail0, detail1;
    // A segment's bytes from literals and the near ring are stored at once; bytes whose source is older than
    // the ring ("far") are requested from HBM/L2 and stored only when the next segment starts (complete_pending),
    // so the far-read latency overlaps the decode of the next windows instead of stalling the wave.
    // Resumable decoders on a small ring (round 4): a call's output goes to THAT call's room, so what is older than the ring
    // is not in `out`: every flushed byte is written a second time, into the decoder's own 32 KiB history in HBM at
    // hist[position & 32767], and far reads come from there (L1-bypassing loads: the history is rewritten as the stream goes).
    static constexpr bool RES_HIST = RES && RING_BITS < 15;
    static constexpr uint32_t HIST_BYTES = 32768u, HIST_MASK = HIST_BYTES - 1u;
    uint8_t *hist;
    const uint8_t *far_base;    // far reads: far_base + 32768 is produced-byte `flushed` (or the input, see set_far_base)
    uint64_t far_okmask;        // all ones while far reads may touch the output (128 <= flushed < cap), else 0
    // What a lane with no far source reads (the far load is unconditional: see segment_store): a byte one whole cache
    // line or more below `flushed` -- or the stream's first input byte while nothing may be read from the output.
    static constexpr uint32_t FAR_IDLE = 32768u - 128u;
    uint64_t pend_m0, pend_m1;  // lanes of the last segment's two 64-byte passes whose byte is still on its way (0 = none)
    uint32_t pend_pos;          // those bytes belong at ring position pend_pos + 64 * pass + lane
    LaneVec<uint8_t> pendF0, pendF1;   // the far bytes (valid in the lanes of pend_m0 / pend_m1); bytes, so that nothing
                                       // (no zero-extension either) touches the loaded registers before complete_pending()
    // extension (PZG_FDICT): a preset dictionary primes the history; only the RING_BITS = 15 instance decodes with one
    const uint8_t *dict;
    uint32_t dict_len;          // 0: none -- a stream with FDICT set then decodes as the reference does (DICTID skipped)
    uint32_t gz_expect;         // gzip: CRC-32 of the whole output according to the member trailers read so far
    uint32_t hist_extra;        // bytes of history in front of the output (min(dict_len, 32768) once the dictionary is installed)
    // resumable instance only
    uint32_t res_final;         // no more input will follow: running out of it is an error, not a suspension
    uint32_t phase, bfinal_cur, stored_left, deferred, ow, chunks;
    uint64_t susp_pos;          // stream bit position (relative to this call's input) at which the next call resumes
    uint64_t in_total_bits;     // 8 * the input bytes consumed by earlier calls
    uint32_t qn;                // tokens waiting in QT (lanes 0..qn-1, never more than QCAP), see queue_append()
    LaneVec<uint32_t> QT;
#if defined(PZG_PROFILE)
    uint64_t prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 0 total, 1 header+tables, 2 token loop, 3 flush, 4 window_append, 5 checked steps, 6 windows, 7 tokens queued, 8-11 emit phases, 12 emit, 13 segments, 14 general copies, 15 checked steps
#endif

    PZG_FN Decoder(WaveLds<RING_BITS> &lds) : L(lds), strip(nullptr) {}

    // Pin every piece of wave-uniform decoder state back into SGPRs.  All of it IS uniform by
    // construction; this only tells the compiler so (v_readfirstlane of an SGPR value folds away),
    // so the state machine is compiled to SALU + s_cbranch instead of VALU + exec-mask loops.
    PZG_FN void pin_uniform()
    {
#if PZG_DEVICE_PASS
        br.rp = uni(br.rp);
        br.rp_ok1 = (int32_t)uni((uint32_t)br.rp_ok1);
        br.rp_ok2 = (int32_t)uni((uint32_t)br.rp_ok2);
        br.chunk0 = uni(br.chunk0);
        br.end_rel = uni64(br.end_rel);
        br.ndw = uni(br.ndw);
        br.mis_bits = uni(br.mis_bits);
        br.win_end = uni(br.win_end);
        op = uni64(op);
        flushed = uni64(flushed);
        adler_a = uni(adler_a);
        adler_b = uni(adler_b);
        lit_e15 = uni(lit_e15);
        dist_e15 = uni(dist_e15);
        lit_n = uni(lit_n);
        use_sub = uni(use_sub);
        lit_sub_used = uni(lit_sub_used);
        dist_sub_used = uni(dist_sub_used);
        s_rd = uni(s_rd);
        s_qn = uni(s_qn);
        s_total = uni(s_total);
        s_cnt = uni(s_cnt);
        s_lc = uni(s_lc);
        dist_n = uni(dist_n);
        pend_m0 = uni64(pend_m0);
        pend_m1 = uni64(pend_m1);
        pend_pos = uni(pend_pos);
        hist_extra = uni(hist_extra);
        far_okmask = uni64(far_okmask);
        qn = uni(qn);
        in_byte0 = uni64(in_byte0);
        status = (int32_t)uni((uint32_t)status);
#endif
    }

    PZG_FN int fail(int32_t st, uint32_t d0, uint32_t d1)
    {
        status = st;
        detail0 = d0;
        detail1 = d1;
        return st;
    }

    // absolute bit offset of the next unread bit within the stream
    PZG_FN uint64_t stream_bit_pos() const { return in_byte0 * 8u + br.pos() - br.mis_bits; }

    // ---- OutputWindow.hs + Adler32.hs: ring -> HBM flush with the checksum folded in ------------
    // Writes produced bytes [flushed, to) and advances the Adler state over them.  `flushed` is
    // always a multiple of 16, so ring offsets and (for a 16-byte aligned output) global
    // addresses are 16-byte aligned: one ds_read_b128 + one global_store_dwordx4 per lane.
    PZG_FN void flush_to(uint64_t to) { flush_span<false>(to); }
    // One KiB from produced-byte p (a multiple of 16) on, whole and aligned: one vector per lane, and every sum fits 32
    // bits -- a lane's byte sum s <= 16 * 255, its weighted sum (1008 - 16 lane) s + sum (16 - i) d_i < 2^23, the wave's
    // < 2^29 -- so nothing is reduced modulo 65521 before the two scalar updates.
    PZG_FN void flush_kib(uint64_t p)
    {
        const uint32_t lane = lane_id();
        uint32_t a_l = 0, b_l = 0;
#pragma nounroll
        for (uint32_t it = 0; it * PZG_WAVE < 64u; ++it) {  // (one round on the device; the one-lane host model takes 64)
            const uint32_t j = it * PZG_WAVE + lane;
            const uint64_t pos = p + (uint64_t)j * 16u;
            const uint32_t roff = (uint32_t)pos & RMASK;
            uint32_t *gp = (uint32_t *)(void *)(out + pos);
#if PZG_DEVICE_PASS
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 rv = *(const u32x4 *)(const void *)&L.ring[roff];  // ds_read_b128
            *(u32x4 *)gp = rv;
            if (RES_HIST) *(u32x4 *)(void *)(hist + ((uint32_t)pos & HIST_MASK)) = rv;
            const uint32_t x0 = rv.x, x1 = rv.y, x2 = rv.z, x3 = rv.w;
#else
            const uint32_t *rp = (const uint32_t *)(const void *)&L.ring[roff];
            const uint32_t x0 = rp[0], x1 = rp[1], x2 = rp[2], x3 = rp[3];
            gp[0] = x0; gp[1] = x1; gp[2] = x2; gp[3] = x3;
            if (RES_HIST) {
                uint32_t *hp = (uint32_t *)(void *)(hist + ((uint32_t)pos & HIST_MASK));
                hp[0] = x0; hp[1] = x1; hp[2] = x2; hp[3] = x3;
            }
#endif
            // Adler32.hs:29-34 advanceNoMod over 16 bytes at once: s = sum d_i, t = sum (16-i) d_i
            const uint32_t s = sum4(x0, sum4(x1, sum4(x2, sum4(x3, 0u))));
            const uint32_t t = dot4(x0, 0x0D0E0F10u, dot4(x1, 0x090A0B0Cu, dot4(x2, 0x05060708u, dot4(x3, 0x01020304u, 0u))));
            a_l += s;
            b_l += (1008u - 16u * j) * s + t;  // weight of byte i of vector j: 1024 - 16 j - i
        }
        const uint32_t sum_a = wave_sum(a_l), sum_b = wave_sum(b_l);
        // Adler32.hs:22-27 in block form: A' = A + sum d ; B' = B + n*A + sum (n - pos) d
        adler_b = (adler_b + 1024u * adler_a + sum_b) % ADLER_MOD;
        adler_a = (adler_a + sum_a) % ADLER_MOD;
    }
    PZG_FN bool out_aligned() const { return (((uintptr_t)out) & 15u) == 0u; }
    // FULL (hot_loop): the span is a whole number of 64-vector rounds (a multiple of 1 KiB), the output is 16-byte aligned
    // and the span ends inside the capacity -- every lane has a whole vector to move in every round, no lane-dependent
    // branch is left.
    template <bool FULL>
    PZG_FN void flush_span(uint64_t to)
    {
        PZG_T0(tf);
        wave_sync();
        const uint64_t from = flushed;
        if (to <= from) return;
        if (FULL) {
            for (uint64_t p = from; p < to; p += 1024u) flush_kib(p);
            flushed = to;
            set_far_base();
            wave_sync();
            PZG_ACC(3, tf);
            return;
        }
        const uint32_t n = (uint32_t)(to - from);
        const uint32_t nvec = (n + 15u) >> 4;
        const uint32_t lane = lane_id();
        const bool out_al = FULL || out_aligned();
#if PZG_DEVICE_PASS
        uint32_t a_l = 0, w_l = 0, u_l = 0;  // a lane sees at most RING / 1024 vectors per flush: no overflow
#else
        uint64_t a_l = 0, w_l = 0, u_l = 0;  // (the one-lane host model walks every vector itself)
#endif
#pragma nounroll
        for (uint32_t it = 0; it * PZG_WAVE < nvec; ++it) {
            const uint32_t j = it * PZG_WAVE + lane;
            if (FULL || j < nvec) {
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
                const uint32_t valid = FULL ? 16u : (to - pos) >= 16u ? 16u : (uint32_t)(to - pos);
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
                if (FULL || (out_al && valid == 16u && pos + 16u <= cap)) {
                    uint32_t *gp = (uint32_t *)(void *)(out + pos);
#if PZG_DEVICE_PASS
                    u32x4 v = {x0, x1, x2, x3};
                    *(u32x4 *)gp = v;
#else
                    gp[0] = x0; gp[1] = x1; gp[2] = x2; gp[3] = x3;
#endif
                } else {
                    const uint32_t xs[4] = {x0, x1, x2, x3};
                    for (uint32_t k = 0; k < 16u; ++k)  // fixed trip count: no lane-dependent loop exit
                        if (k < valid && pos + k < cap) out[pos + k] = (uint8_t)(xs[k >> 2] >> (8u * (k & 3u)));
                }
                if (RES_HIST) {  // ... and into the decoder's history (whole vectors: `pos` is a multiple of 16, zeros past `to` are rewritten by the flush that follows)
                    uint32_t *hp = (uint32_t *)(void *)(hist + ((uint32_t)pos & HIST_MASK));
#if PZG_DEVICE_PASS
                    u32x4 hv = {x0, x1, x2, x3};
                    *(u32x4 *)hp = hv;
#else
                    hp[0] = x0; hp[1] = x1; hp[2] = x2; hp[3] = x3;
#endif
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
        uint32_t bl_mod;
        if (PZG_WAVE == 64u && RING_BITS <= 13) {
            // (device, small rings: n <= 8 KiB, a lane sees at most 8 vectors, so every term is below 2^29 and the sum -- a
            // sum of positive weights times bytes -- is not negative: 32-bit arithmetic, a third of the registers)
            const uint32_t bl32 = (n - 16u * lane - 16u) * (uint32_t)a_l + (uint32_t)w_l - (16u * PZG_WAVE) * (uint32_t)u_l;
            bl_mod = bl32 % ADLER_MOD;
        } else {
            int64_t bl = (int64_t)((int64_t)n - 16 * (int64_t)lane - 16) * (int64_t)a_l + (int64_t)w_l -
                         (int64_t)(16u * PZG_WAVE) * (int64_t)u_l;
            bl_mod = (uint32_t)((uint64_t)bl % ADLER_MOD);
        }
        uint32_t sum_a = wave_sum((uint32_t)(a_l % ADLER_MOD));
        uint32_t sum_b = wave_sum(bl_mod);
        // Adler32.hs:22-27 in block form: A' = A + sum d ; B' = B + n*A + sum (n - pos) d
        uint64_t nb = (uint64_t)adler_b + (uint64_t)(n % ADLER_MOD) * adler_a + sum_b;
        adler_a = (uint32_t)(((uint64_t)adler_a + sum_a) % ADLER_MOD);
        adler_b = (uint32_t)(nb % ADLER_MOD);
        flushed = to;
        set_far_base();
        wave_sync();
        PZG_ACC(3, tf);
    }
    // Far sources lie below `flushed`, so inside the capacity while `flushed` is; a stream that has outgrown its capacity
    // is redone by the 32 KiB-ring kernel anyway (its far bytes were never stored): it reads its input instead.
    // Recomputed per flush (every KiB or so), not per segment.
    // Round 3: far reads are plain (cached) loads -- 25 % fewer bytes fetched from beyond the L2 than with `nt`, same speed --
    // which is safe because every line a far read touches is COMPLETE and final when it is read: a far source lies more than
    // 1 KiB below `flushed` (op - flushed <= RING - 1024), only the line AT `flushed` can be partly written, output bytes
    // are written once, and the idle lanes read FAR_IDLE (a whole line below `flushed`; the input before anything is
    // flushed), never the line the next flush will write.  So no cache can hold a copy that a later store outdates.
    PZG_FN void set_far_base()
    {
        if (!HYBRID) return;
        if (RES_HIST) {  // (a resumable decoder never produces past its room: everything older than the ring is in the history)
            far_base = hist;
            far_okmask = ~0ull;
            return;
        }
        const bool ok = flushed < cap && flushed >= 128u;  // (below 2 KiB of output nothing is far anyway)
        far_base = ok ? out + (flushed - 32768u) : in - FAR_IDLE;
        far_okmask = ok ? ~0ull : 0ull;
    }

    // store the last segment's far bytes (see pend_m0), then flush if due: from here on every byte below `op` is in the ring
    PZG_FN void complete_pending()
    {
        pending_stores();
        maybe_flush();
    }
    PZG_FN void pending_stores()
    {
        if ((pend_m0 | pend_m1) != 0ull) {
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
            {
                PZG_T0(t_w);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PZG_ACC(10, t_w);  // how long the far bytes are waited for
            }
#endif
            if (pend_m0 != 0ull) {
                PZG_LANES_BEGIN(j)
                    ring_store(lane_bit(pend_m0, j), ((pend_pos + j)) & RMASK, PZG_LV(pendF0, j), j);
                PZG_LANES_END
            }
            if (pend_m1 != 0ull) {
                PZG_LANES_BEGIN(j)
                    ring_store(lane_bit(pend_m1, j), ((pend_pos + 64u + j)) & RMASK, PZG_LV(pendF1, j), j);
                PZG_LANES_END
            }
            pend_m0 = pend_m1 = 0ull;
        }
    }

    PZG_FN void maybe_flush()
    {
        if ((uint32_t)(op - flushed) >= FLUSH_AT) flush_to(op & ~(uint64_t)15u);  // the difference never exceeds RING
    }

    // Lane-predicated LDS byte store WITHOUT a branch: lanes whose predicate is false store into a
    // per-lane dump byte instead.  A lane-dependent branch inside the loops that carry the
    // wave-uniform decoder state makes hipcc treat that state as divergent (VALU + exec-mask loops
    // instead of SALU + s_cbranch); a select on the address does not.
    PZG_FN void sel_store(bool pred, uint8_t *dst, uint8_t v, uint32_t lane)
    {
        uint8_t *p = pred ? dst : &L.dump[lane & 63u];
        *p = v;
    }
    // the same into the ring at index `idx` (already reduced modulo RING): the select is made on the offset from the
    // ring's base, so the base itself rides in the store's immediate offset (one vector instruction less per store)
    PZG_FN void ring_store(bool pred, uint32_t idx, uint8_t v, uint32_t lane)
    {
        constexpr uint32_t DUMP_REL = (uint32_t)(offsetof(WaveLds<RING_BITS>, dump) - offsetof(WaveLds<RING_BITS>, ring));
        uint8_t *base = L.ring;
        base[pred ? idx : DUMP_REL + lane] = v;  // lane < 64 (the dump has room for 76)
    }

    // The byte `back` positions before the output cursor (1 <= back <= 32768, back <= op).
    // The LDS ring holds the last RING bytes.  With RING_BITS == 15 that is the whole DEFLATE window.
    // With a smaller ring (more resident stream-waves per CU) older bytes come from the stream's own
    // output in HBM/L2: everything older than the ring has been flushed (op - flushed <= FLUSH_AT) and
    // far_fence() waits for the flush's stores; the lines read are complete and final by then (set_far_base), so the
    // segments' far loads are plain cached loads (this rare path keeps round 2's non-temporal ones).
    static constexpr bool HYBRID = RING_BITS < 15;
    PZG_FN uint8_t fetch_near(uint32_t back) const { return L.ring[((uint32_t)op - back) & RMASK]; }
    // Before far reads: every flush store of this wave must have landed.  By the time a byte is older
    // than the ring its flush is long complete, so this wait is normally free.
    PZG_FN void far_fence() const
    {
#if PZG_DEVICE_PASS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    PZG_FN uint8_t fetch_far(bool is_far, uint32_t back) const
    {
        // lanes that are not far (or a count-only stream whose bytes were never stored) read byte 0 of the output
        uint64_t p = op - back;
        if (RES_HIST) {
            const uint32_t hp = is_far ? (uint32_t)p & HIST_MASK : 0u;
#if PZG_DEVICE_PASS
            return __builtin_nontemporal_load(hist + hp);
#else
            return hist[hp];
#endif
        }
        const bool ok = is_far && p < cap;
        p = ok ? p : 0u;
#if PZG_DEVICE_PASS
        const uint8_t v = cap ? __builtin_nontemporal_load(out + p) : (uint8_t)0;
#else
        const uint8_t v = cap ? out[p] : (uint8_t)0;
#endif
        return v;
    }

    // Monad.hs:309-315 emitByte -> OutputWindow.hs:64-68 addByte
    PZG_FN void put_literal(uint32_t v)
    {
        const uint32_t lane = lane_id();
        ring_store(lane == 0u, ((uint32_t)op) & RMASK, (uint8_t)v, lane);
        op++;
    }

    // Monad.hs:324-333 emitPastChunk -> OutputWindow.hs:82-101 addOldChunk/copyChunked.
    // Lane-cooperative LZ77 copy.  Every source byte lies before `op`: for dist >= len it is
    // op-dist+k, for dist < len the pattern repeats with period dist (copyChunked's dist-sized
    // pieces), i.e. op-dist+(k mod dist).  So all reads are issued before any write.
    PZG_FN void copy_match(uint32_t dist, uint32_t len)
    {
        const uint32_t lane = lane_id();
        const uint32_t src0 = (uint32_t)op - dist;
        const uint32_t dst0 = (uint32_t)op;
        if (len <= PZG_WAVE && dist >= len) {  // the common case: one read, one write
            uint8_t v = L.ring[(src0 + lane) & RMASK];  // lanes >= len read a harmless ring byte
            if (HYBRID) {
                const bool far = lane < len && dist - lane > RING;
                if (ballot(far)) {
                    far_fence();
                    v = far ? fetch_far(far, dist - lane) : v;
                }
            }
            ring_store(lane < len, ((dst0 + lane)) & RMASK, v, lane);
            op += len;
            return;
        }
        constexpr uint32_t MAXCH = (258u + PZG_WAVE - 1u) / PZG_WAVE;
        uint8_t v[MAXCH];
        const bool overlap = dist < len;
#if PZG_DEVICE_PASS
        const float rd = overlap ? __builtin_amdgcn_rcpf((float)dist) : 0.0f;
        uint32_t lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // (opaque: or lane + 64 c + 0.5 of every chunk is computed once per kernel and kept, or spilled)
#else
        const uint32_t lane_o = lane;
#endif
#pragma unroll
        for (uint32_t c = 0; c < MAXCH; ++c) {
            const uint32_t k = c * PZG_WAVE + lane_o;
            if (c * PZG_WAVE < len) {  // wave-uniform
                uint32_t off = k;
                if (overlap) {
#if PZG_DEVICE_PASS
                    uint32_t q = (uint32_t)(((float)k + 0.5f) * rd);
                    off = k - q * dist;
                    off = off >= dist ? off - dist : off;
#else
                    off = k % dist;
#endif
                }
                v[c] = L.ring[(src0 + (k < len ? off : 0u)) & RMASK];
                if (HYBRID) {  // only a non-overlapping match (off == k) can reach past the ring
                    const bool far = k < len && dist - off > RING;
                    if (ballot(far)) {
                        far_fence();
                        v[c] = far ? fetch_far(far, dist - off) : v[c];
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < MAXCH; ++c) {
            const uint32_t k = c * PZG_WAVE + lane;
            if (c * PZG_WAVE < len) ring_store(k < len, ((dst0 + k)) & RMASK, v[c], lane);
        }
        op += len;
    }

    // ---- Deflate.hs:255-292 + HuffmanTree.hs: canonical code -> two-level table -----------------
    // lens[0..n) in LDS.  Builds the 2^P primary LUT, the sorted-symbol permutation and the
    // per-length first/count/offset table.  Returns false when the code is over-subscribed,
    // which is exactly when createHuffmanTree returns Left (any overlap of canonical codes).
    // how many bits past its P-bit prefix c_p the longest code with that prefix has (0: none / lane not on), capped at
    // SUB_DEPTH_MAX: the codes of length l are the canonical range [first[l], first[l] + count[l]) at that length
    static constexpr uint32_t SUB_DEPTH_MAX = 7u;
#ifndef PZG_SUB_LONG_RATE
#define PZG_SUB_LONG_RATE 16
#endif
    static constexpr uint32_t SUB_LONG_RATE = PZG_SUB_LONG_RATE;  // (units of 2^-15: 16 = one token in 2048)

    // (metav: lane l holds first[l] << 16 | count[l] -- a register, so that no LDS round trip is paid per length)
    template <int P>
    PZG_FN uint32_t sub_depth(const TreeMeta *meta, uint32_t metav, uint32_t c_p, bool on)
    {
        uint32_t depth = 0;
#pragma nounroll
        for (uint32_t l = (uint32_t)P + 1u; l < 16u; ++l) {
#if PZG_DEVICE_PASS
            (void)meta;
            const uint32_t fc = read_lane(metav, l), cnt_l = fc & 0xffffu, first_l = fc >> 16;
#else
            (void)metav;
            const uint32_t cnt_l = meta->count[l], first_l = meta->first[l];
#endif
            const uint32_t sh = l - (uint32_t)P;
            const bool hit = on && cnt_l != 0u && c_p >= (first_l >> sh) && c_p <= ((first_l + cnt_l - 1u) >> sh);
            depth = hit ? (sh < SUB_DEPTH_MAX ? sh : SUB_DEPTH_MAX) : depth;
        }
        return depth;
    }

    template <int P, int TREE>
    PZG_FN bool build_table(const uint8_t *lens, uint32_t n, uint32_t *lut, TreeMeta *meta, uint32_t *e15_out)
    {
        const uint32_t lane = lane_id();
        // pass 1: histogram of code lengths (blCount, Deflate.hs:266)
        wave_sync();
        if (lane < 16u) L.cnt[lane] = 0u;
        if (PZG_WAVE == 1u)
            for (uint32_t i = 1; i < 16u; ++i) L.cnt[i] = 0u;
        wave_sync();
#pragma nounroll
        for (uint32_t s0 = 0; s0 < n; s0 += PZG_WAVE) {  // every loop here has a wave-uniform trip count
            const uint32_t s = s0 + lane;
            const uint32_t len = s < n ? lens[s] : 0u;
#if PZG_DEVICE_PASS
            if (len) atomicAdd(&L.cnt[len], 1u);
#else
            if (len) L.cnt[len]++;
#endif
        }
        wave_sync();
        // next_code (step2, Deflate.hs:273-278), offsets, Kraft sum; wave-uniform, kept in LDS (meta)
        uint32_t code = 0, prev = 0, e15 = 0, covered_p = 0, maxlen = 0;
        uint32_t metav = 0;  // lane l: first[l] << 16 | count[l] (see sub_depth)
#pragma nounroll
        for (uint32_t l = 1; l < 16u; ++l) {
            const uint32_t c = uni(L.cnt[l]);
            if (c) maxlen = l;
            code = (code + prev) << 1;
            prev = c;
            metav = lane == l ? ((code & 0xffffu) << 16) | c : metav;
            if (lane == 0u) {
                meta->count[l] = (uint16_t)c;
                meta->first[l] = (uint16_t)code;
                L.cnt[l] = 0u;  // becomes the running rank of pass 2
            }
            e15 += c << (15u - l);
            if (l == (uint32_t)P) covered_p = code + c;  // P-bit prefixes covered by codes of length <= P: [0, covered_p)
        }
        *e15_out = e15;
        if (e15 > 32768u) return false;  // over-subscribed: some insertion must collide
        // Second level (literal/length and distance).  The P-bit prefixes of the codes longer than P are
        // the contiguous canonical range [covered_p, end_p); each of the first np_fit of them gets a table
        // of 2^sb entries at sub[sub0 + ((c_p - covered_p) << sb)], indexed by the next sb stream bits.
        // The literal/length tables come first in the pool, the distance tables take what is left.
        // token_step_checked() always resolves through them; the windows do their second lookup only
        // when the literal/length code has enough long prefixes for it to pay (use_sub).  Whatever the
        // tables do not resolve (longer codes, holes of an incomplete code) stays K_LONG and goes the
        // exact way: decode_long().
        // Two ways to share the pool.  UNIFORM: every long prefix gets 2^sb entries, sb the largest that fits (cheap to set
        // up: the K_SUB entries are arithmetic).  When that leaves more than SUB_LONG_RATE of the tokens to decode_long() --
        // a code of length l is used once in 2^l tokens, so the fraction is a sum over the histogram -- PER PREFIX: each
        // prefix gets a table as deep as ITS longest code needs; the canonical order puts the longest codes behind the
        // fewest prefixes, so the pool goes much further (literal-heavy data: 3x fewer checked steps, +33 %), at ~1,400
        // more scalar instructions per header (text would lose 1 % to them: it keeps the uniform tables).  If the depths
        // do not fit they are capped at the largest value that does (dcap); if two entries per prefix do not fit either,
        // only the first np_fit prefixes get a table.
        uint32_t np_fit = 0, sub0 = 0, dcap = 0, dup = 0, sub_total = 0, sb = 0;
        bool uniform = true;
        if (TREE != TREE_CODELEN) {
            sub0 = TREE == TREE_DIST ? lit_sub_used : 0u;
            const uint32_t pool = SUB_ENTRIES - sub0;
            uint32_t end_p = (e15 + (1u << (15u - P)) - 1u) >> (15u - P);
            if (end_p > (1u << P)) end_p = 1u << P;
            const uint32_t np = end_p > covered_p ? end_p - covered_p : 0u;
            if (maxlen > (uint32_t)P && np != 0u && pool >= 2u) {
                sb = maxlen - (uint32_t)P < SUB_BITS_MAX ? maxlen - (uint32_t)P : SUB_BITS_MAX;
                while (sb > 1u && (np << sb) > pool) --sb;
                np_fit = (pool >> sb) < np ? (pool >> sb) : np;
                sub_total = np_fit << sb;
                // what the uniform tables leave unresolved, in units of 2^-15 of the tokens (prefixes without a table count
                // as all of it: that case always goes per prefix)
                uint32_t beyond = np_fit < np ? ~0u : 0u;
#pragma nounroll
                for (uint32_t l = 15u; l > (uint32_t)P + sb && beyond <= SUB_LONG_RATE; --l) {
#if PZG_DEVICE_PASS
                    const uint32_t cnt_l = read_lane(metav, l) & 0xffffu;
#else
                    const uint32_t cnt_l = meta->count[l];
#endif
                    beyond += cnt_l << (15u - l);
                }
                uniform = beyond <= SUB_LONG_RATE;
            }
            if (!uniform) {
                // prefixes per depth (prefix q's depth: the largest l - P whose canonical range of length-l codes reaches it)
                uint32_t per_depth[SUB_DEPTH_MAX + 1u];
                for (uint32_t d = 0; d <= SUB_DEPTH_MAX; ++d) per_depth[d] = 0u;
#pragma nounroll
                for (uint32_t q0 = 0; q0 < np; q0 += PZG_WAVE) {
                    const uint32_t c_p = covered_p + q0 + lane;
                    const uint32_t dq = sub_depth<P>(meta, metav, c_p, q0 + lane < np);
                    if (q0 + lane < np) lut[bitrev32(c_p) >> (32u - P)] = dq;  // (kept there until the K_SUB entries are written below)
                    per_depth[0] += popc64(ballot(q0 + lane < np && dq == 0u));  // (a hole of an incomplete code: one K_LONG entry)
#pragma nounroll
                    for (uint32_t d = 1; d <= SUB_DEPTH_MAX; ++d) per_depth[d] += popc64(ballot(dq == d));
                }
                dcap = SUB_DEPTH_MAX;
                for (;;) {
                    sub_total = per_depth[0];
                    for (uint32_t d = 1; d <= SUB_DEPTH_MAX; ++d) sub_total += per_depth[d] << (d < dcap ? d : dcap);
                    if (sub_total <= pool || dcap == 1u) break;
                    --dcap;
                }
                np_fit = np;
                if (sub_total > pool) {  // (dcap == 1: two entries per prefix)
                    np_fit = pool >> 1;
                    sub_total = np_fit << 1;
                } else if (dcap < SUB_DEPTH_MAX) {
                    // (round 6) What the cap leaves of the pool goes to the first prefixes it cut short, one level each: a code that
                    // needed 254 entries of 252 lost ALL its 10-bit symbols to decode_long() -- binary-looking records, 256 literals of
                    // 8 to 10 bits: 900 checked steps per 30 KiB stream, every span ended by one -- where it now loses a few.
                    uint32_t deep = 0u;
                    for (uint32_t d = dcap + 1u; d <= SUB_DEPTH_MAX; ++d) deep += per_depth[d];
                    dup = (pool - sub_total) >> dcap;
                    if (dup > deep) dup = deep;
                    sub_total += dup << dcap;
                }
            }
            if (TREE == TREE_DIST) dist_sub_used = sub_total;
            if (TREE == TREE_LITLEN) {
                lit_sub_used = sub_total;
                use_sub = np_fit >= SUB_MIN_PREFIXES ? (uniform ? 1u : 3u) : 0u;  // (bit 1: tables per prefix -- long codes are in constant use: strip_span resolves them)
#if defined(PZG_STATS) && defined(PZG_STATS_NOSUB) && !PZG_DEVICE_PASS
                use_sub = 0u;  // (lab statistics: how many tokens need the second level at all)
#endif
            }
#pragma nounroll
            for (uint32_t i0 = 0; i0 < sub_total; i0 += PZG_WAVE) {
                const uint32_t i = i0 + lane;
                if (i < sub_total) L.sub[sub0 + i] = mk_stop(0, K_LONG, 0);
            }
        }
        wave_sync();
        // default fill: patterns no code of length <= P covers are either the prefix of a longer
        // code (K_LONG) or lead the reference's trie walk into HuffmanEmpty at some depth d
        // (HuffmanTree.hs:78-80): the first d whose d-bit prefix lies at or past the end of all codes.
        if (covered_p < (1u << P)) {
#pragma nounroll
            for (uint32_t i0 = 0; i0 < (1u << P); i0 += PZG_WAVE) {
                const uint32_t idx = i0 + lane;
                const uint32_t c_p = bitrev32(idx) >> (32u - P);  // MSB-first value of the P stream bits
                uint32_t ent;
                if (e15 == 0u) {
                    ent = mk_stop(1, K_EMPTY_TREE, 0);
                } else if ((c_p << (15u - P)) < e15) {
                    ent = (TREE != TREE_CODELEN && uniform && c_p - covered_p < np_fit) ? (mk_stop(sb, K_SUB, sub0 + ((c_p - covered_p) << sb)) | ENT_SUB)
                                                                                      : mk_stop(0, K_LONG, 0);
                } else {
                    uint32_t d = (uint32_t)P;  // smallest d whose d-bit prefix is at or past the end of all codes
#pragma unroll
                    for (uint32_t t = (uint32_t)P - 1u; t >= 1u; --t)
                        if (((c_p >> ((uint32_t)P - t)) << (15u - t)) >= e15) d = t;
                    ent = mk_stop(d, K_EMPTY_BRANCH, 0);
                }
                // (the prefixes that get a second-level table hold their depth for the moment: not touched here)
                if (idx < (1u << P) && c_p >= covered_p && !(TREE != TREE_CODELEN && !uniform && c_p - covered_p < np_fit)) lut[idx] = ent;
            }
        }
        if (TREE != TREE_CODELEN && !uniform && np_fit != 0u) {  // the K_SUB entries, in canonical order of the prefixes (offsets = running sum of the table sizes)
            uint32_t run = 0, ncut = 0;
#pragma nounroll
            for (uint32_t q0 = 0; q0 < np_fit; q0 += PZG_WAVE) {
                const uint32_t q = q0 + lane, c_p = covered_p + q;
                const uint32_t idx = bitrev32(c_p) >> (32u - P);
                const uint32_t dq = lut[q < np_fit ? idx : 0u];  // its depth, left there by the counting pass
                const bool cut = q < np_fit && dq > dcap;        // ... deeper than the cap: the first `dup` of these get one level more
                const uint64_t cutm = ballot(cut);
                const uint32_t depth = dq < dcap ? dq : (cut && ncut + mbcnt(cutm) < dup ? dcap + 1u : dcap);
                ncut += popc64(cutm);
                const uint32_t size = q < np_fit ? 1u << depth : 0u;
                const uint32_t incl = wave_iscan_add(size);
                if (q < np_fit) lut[idx] = mk_stop(depth, K_SUB, sub0 + run + incl - size) | ENT_SUB;
#if PZG_DEVICE_PASS
                run += read_lane(incl, 63u);
#else
                run += incl;
#endif
            }
        }
        wave_sync();
        // pass 2: canonical code of every symbol (step3, Deflate.hs:280-288): first[len] + rank among
        // the symbols of equal length below it; then the replicated LUT fill.  Ranks come from
        // ballots over the lengths present in each 64-symbol round, in symbol order.
#pragma nounroll
        for (uint32_t s0 = 0; s0 < n; s0 += PZG_WAVE) {
            const uint32_t s = s0 + lane;
            const uint32_t len = s < n ? lens[s] : 0u;
            uint32_t rank = 0;
            uint64_t remaining = ballot(len != 0u);
#pragma nounroll
            while (remaining) {
                const uint32_t lsel = read_lane(len, ctz64(remaining));
                const bool mine = len == lsel;
                const uint64_t m = ballot(mine);
                const uint32_t basec = uni(L.cnt[lsel]);
                if (mine) rank = basec + mbcnt(m);
                if (lane == 0u) L.cnt[lsel] = basec + popc64(m);
                remaining &= ~m;
            }
            if (len != 0u) {
                const uint32_t c = (uint32_t)meta->first[len] + rank;
                if (len <= (uint32_t)P) {
                    uint32_t ent = TREE == TREE_LITLEN ? litlen_entry(s, len)
                                   : TREE == TREE_DIST ? dist_entry(s, len)
                                                       : codelen_entry(s, len);
                    const uint32_t rev = bitrev32(c) >> (32u - len);
#pragma nounroll
                    for (uint32_t idx = rev; idx < (1u << P); idx += (1u << len)) lut[idx] = ent;
                } else if (TREE != TREE_CODELEN && np_fit != 0u) {
                    const uint32_t rl = len - (uint32_t)P;   // bits of the code past its P-bit prefix
                    const uint32_t pfx = (c >> rl) - covered_p;
                    if (pfx < np_fit) {
                        const uint32_t ps = lut[bitrev32(c >> rl) >> (32u - P)];  // the prefix' K_SUB entry: its table's depth and place
                        const uint32_t depth = ent_n(ps);
                        if (rl <= depth) {
                            const uint32_t ent = TREE == TREE_LITLEN ? litlen_entry(s, len) : dist_entry(s, len);
                            const uint32_t rev = bitrev32(c & ((1u << rl) - 1u)) >> (32u - rl);  // those bits in stream order
#pragma nounroll
                            for (uint32_t idx = rev; idx < (1u << depth); idx += (1u << rl)) L.sub[ent_sub_index(ps) + idx] = ent;
                        }
                    }
                }
            }
        }
        wave_sync();
        return true;
    }

    // Second level (codes longer than the primary table): canonical first-code walk, one length
    // per step, equivalent to HuffmanTree.hs:73-83 advanceTree on the same bits.  The symbol with
    // canonical index `idx` among those of length l is found by a ballot scan of the block's code
    // lengths, which stay in LDS for the whole block (no sorted-symbol table: LDS is what bounds the
    // number of resident stream-waves).  Returns an entry.
    template <int TREE>
    PZG_FN uint32_t decode_long(uint32_t bits, const TreeMeta *meta, const uint8_t *lens, uint32_t nsym, uint32_t e15)
    {
        // The primary table sent us here (K_LONG), so no code of length <= P matches and the P-bit prefix
        // is below the end of all codes: the walk resumes at depth P + 1.
        constexpr uint32_t P = TREE == TREE_LITLEN ? (uint32_t)LIT_BITS : (uint32_t)DIST_BITS;
        uint32_t code = bitrev32(bits) >> (32u - P);
        bits >>= P;
#pragma nounroll
        for (uint32_t l = P + 1u; l < 16u; ++l) {
            code = (code << 1) | (bits & 1u);
            bits >>= 1;
            const uint32_t cnt_l = uni(meta->count[l]);
            const uint32_t first_l = uni(meta->first[l]);
            if (code - first_l < cnt_l && code >= first_l) {
                uint32_t want = code - first_l, sym = 0;
                const uint32_t lane = lane_id();
#pragma nounroll
                for (uint32_t s0 = 0; s0 < nsym; s0 += PZG_WAVE) {
                    const uint32_t s = s0 + lane;
                    const bool mine = s < nsym && lens[s < nsym ? s : 0u] == l;
                    const uint64_t m = ballot(mine);
                    const uint32_t c = popc64(m);
                    if (want < c) {
                        const uint64_t hit = ballot(mine && mbcnt(m) == want);
                        sym = s0 + ctz64(hit);
                        break;
                    }
                    want -= c;
                }
                return TREE == TREE_LITLEN ? litlen_entry(sym, l) : dist_entry(sym, l);
            }
            if ((code << (15u - l)) >= e15) return mk_stop(l, K_EMPTY_BRANCH, 0);
        }
        return mk_stop(15, K_EMPTY_BRANCH, 0);  // unreachable: e15 <= 2^15 ends every walk by 15
    }

    // Checks a non-symbol entry against the real bits left, in the reference's order: the walk
    // runs out of data (TRUNCATED) before any error that needs a later bit.
    PZG_FN int check_entry(uint32_t ent)
    {
        const uint32_t kind = ent_is_stop(ent) ? ent_stop_kind(ent) : (uint32_t)K_LIT;
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

    // ---- the wave's token queue (see window_append / emit_segment below) ---------------------------
    // A queued token is one dword:  match  TK_MATCH | len << 16 | dist      literal  1 << 16 | byte << 8 | (junk in [7:0])
    // -- a literal's token is its LUT entry as it stands, so the windows spend no instruction on it.
    static constexpr uint32_t TK_MATCH = ENT_MATCH;
    static constexpr uint32_t QCAP = 63u;   // queue lanes 0..62; lane 63 receives what the compaction discards
#ifndef PZG_QHIGH
#define PZG_QHIGH 40
#endif
    static constexpr uint32_t QHIGH = PZG_QHIGH;  // emit when this many tokens wait (room for any ordinary window stays)

    PZG_FN void queue_push(uint32_t tk)  // qn < QCAP
    {
#if !PZG_DEVICE_PASS
        if (qn >= QCAP) __builtin_trap();  // (host model: the queue never holds more than QCAP tokens -- emit_segment's masks rely on it)
#endif
        PZG_LANES_BEGIN(j)
            PZG_LV(QT, j) = j == qn ? tk : PZG_LV(QT, j);
        PZG_LANES_END
        qn += 1u;
    }

    // ---- table geometry --------------------------------------------------------------------------------------------
    // Dynamic blocks: 2^8-entry primary tables + second-level tables (above).  Fixed-Huffman blocks (FX; not in the
    // resumable instance) get tables of their own shape: the fixed literal/length code is at most 9 bits, so a 2^9-entry
    // table over the two primary tables' LDS resolves every code in ONE lookup (its 9-bit literals 144..255 would
    // otherwise all take the second-level path), and the 5-bit distance code takes a 2^5-entry table in the
    // second-level pool.  Same LDS, no second lookups, no long codes.
    static constexpr bool FX_TABLES = !RES;
    template <bool FX> static constexpr uint32_t lit_bits() { return FX ? 9u : (uint32_t)LIT_BITS; }
    template <bool FX> static constexpr uint32_t dist_bits() { return FX ? 5u : (uint32_t)DIST_BITS; }
    template <bool FX> PZG_FN const uint32_t *lit_table() const { return L.lit_lut; }  // (FX: runs on into dist_lut, its neighbour)
    template <bool FX> PZG_FN const uint32_t *dist_table() const { return FX ? L.sub : L.dist_lut; }

    // ---- Deflate.hs:106-120 runInflate: one token, every bit checked against the stream end ------
    // Used for whatever a window does not handle itself
    // (end-of-block, codes longer than the primary tables, every error).  A literal or match is put
    // on the token queue like the windows' tokens.
    // Returns ST_OK (token consumed), 1000 (end of block consumed) or an error status.
    static constexpr int STEP_EOB = 1000;
    template <bool FX>
    PZG_FN int token_step_checked()
    {
        uint32_t bits = br.peek32();
        uint32_t e = uni(lit_table<FX>()[bits & ((1u << lit_bits<FX>()) - 1u)]);
        uint32_t kind = ent_kind_lit(e);
        if (kind == K_SUB) {  // second level: one more (wave-uniform) lookup
            e = uni(L.sub[ent_sub_index(e) + ((bits >> LIT_BITS) & ((1u << ent_n(e)) - 1u))]);
            kind = ent_kind_lit(e);
        }
        if (kind == K_LONG) {  // the exact walk for whatever the tables do not hold
            e = decode_long<TREE_LITLEN>(bits, &L.lit_meta, L.lens, lit_n, lit_e15);
            kind = ent_kind_lit(e);
        }
        if (kind == K_LIT) {
            const uint32_t n = ent_n(e);
            if (br.avail() < (int64_t)n) return fail(ST_TRUNCATED, 0, 0);
            br.drop(n);
            queue_push(e);
            return ST_OK;
        }
        if (kind == K_BASE) {
            const uint32_t n = ent_base_n(e), tot = ent_base_tot(e);
            if (br.avail() < (int64_t)tot) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t len = ent_len_base(e) + ((bits >> n) & ((1u << (tot - n)) - 1u));
            br.drop(tot);
            bits = br.peek32();
            uint32_t d = uni(dist_table<FX>()[bits & ((1u << dist_bits<FX>()) - 1u)]);
            uint32_t dk = ent_kind_dist(d);
            if (dk == K_SUB) {
                d = uni(L.sub[ent_sub_index(d) + ((bits >> DIST_BITS) & ((1u << ent_n(d)) - 1u))]);
                dk = ent_kind_dist(d);
            }
            if (dk == K_LONG) {
                d = decode_long<TREE_DIST>(bits, &L.dist_meta, L.lens + lit_n, dist_n, dist_e15);
                dk = ent_kind_dist(d);
            }
            if (dk != K_BASE) {
                if (int st = check_entry(d)) return st;
                // K_BADSYM: distanceArray ! c out of range, the reference throws (Deflate.hs:199-205)
                return fail(ST_BAD_DIST_SYMBOL, ent_val(d), 0);
            }
            const uint32_t dn = ent_base_n(d), dtot = ent_base_tot(d);
            if (br.avail() < (int64_t)dtot) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t dist = ent_val(d) + ((bits >> dn) & ((1u << (dtot - dn)) - 1u));
            br.drop(dtot);
            queue_push(TK_MATCH | (len << 16) | dist);  // emit_segment() checks the distance against what has been produced by then
            return ST_OK;
        }
        if (int st = check_entry(e)) return st;
        if (kind == K_EOB) {
            br.drop(ent_n(e));
            return STEP_EOB;
        }
        // K_BADSYM: lengthArray ! c out of range, the reference throws (Deflate.hs:160-166)
        return fail(ST_BAD_LITLEN_SYMBOL, ent_val(e), 0);
    }

    // ---- Deflate.hs:106-120 runInflate, wave-parallel ---------------------------------------------
    // Three cooperating pieces:
    //   window2_decode() lane k decodes the tokens that would start k and k + 64 bits ahead of the cursor
    //                    (literal/length lookup, length extra bits, distance lookup at its own offset,
    //                    distance extra bits); a scalar walk visits the offsets that really are token
    //                    starts; those tokens are compacted onto the tail of the wave's token queue.
    //   emit_body()      takes tokens worth <= 128 output bytes from the head of the queue and produces
    //                    their bytes in two passes of one ring gather and one ring store each.
    //   hot_loop() / token_loop()  keep the queue deep enough that a segment is (nearly) always full.
    //                    token_step_checked() decodes what a window cannot (long codes, end of block, errors)
    //                    onto the same queue; at an end of block or an error the queue is drained
    //                    first, so errors surface in stream order, as in the reference.

    // One lane's speculative decode: the token whose first bit is bit r of (hi:mid:lo).
    // tb = its length in bits, 64 if it is not a plain literal/match (the walk stops there); tk = the token.
    // The three stages of one lane's speculative decode, split so that a 128-bit window can run its two
    // decodes in lockstep (both LDS lookups of a stage are in flight together).
    // When the block's literal/length table has second-level tables (use_sub), long codes are looked up there.
    struct Spec {
        uint32_t w_lo, w_hi, e, w2, d;
    };
    template <bool FX>
    PZG_FN void spec_bits(Spec &t, uint32_t lo, uint32_t mid, uint32_t hi, uint32_t r)
    {
        t.w_lo = funnel(mid, lo, r);  // 64 stream bits from the token's first
        t.w_hi = funnel(hi, mid, r);
        t.e = lit_table<FX>()[t.w_lo & ((1u << lit_bits<FX>()) - 1u)];
    }
    // Second-level lookup in five vector instructions and with NO lane-dependent branch (one would send the whole token
    // loop's control flow through the compiler's structurizer: see hot_loop): every lane looks up -- where its entry is
    // no K_SUB, at an offset that means nothing: inside the wave's LDS that is some other word, past it the hardware
    // returns 0 -- and keeps its own entry.  (A K_SUB entry's bits 14, 15 are clear: (e >> 14) & 0x3fc is its table's
    // byte offset in the pool.)
    PZG_FN uint32_t spec_sub_load(const Spec &t)
    {
        const uint32_t off = ((t.e >> 14) & 0x3fcu) + (ubfe(t.w_lo, LIT_BITS, t.e) << 2);
#if PZG_DEVICE_PASS
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(L.sub) + off);
#else
        return (int32_t)t.e >= (int32_t)ENT_SUB ? L.sub[off >> 2] : 0u;
#endif
    }
    PZG_FN void spec_sub_pick(Spec &t, uint32_t e2)
    {
        t.e = (int32_t)t.e >= (int32_t)ENT_SUB ? e2 : t.e;  // bit 30 set, bit 31 clear: one compare
    }
    PZG_FN void spec_sub(Spec &t)
    {
        uint32_t e2 = spec_sub_load(t);
#if PZG_DEVICE_PASS
        asm("" : "+v"(e2));  // (a load that only feeds a select gets a branch around it otherwise)
#endif
        spec_sub_pick(t, e2);
    }
    template <bool FX>
    PZG_FN void spec_dist(Spec &t)
    {
        t.w2 = funnel(t.w_hi, t.w_lo, t.e);  // bits after the length code and its extra bits: the entry's [4:0], <= 20
        t.d = dist_table<FX>()[t.w2 & ((1u << dist_bits<FX>()) - 1u)];
    }
    // tb = the token's length in bits, >= 128 if it is not a plain literal/match (the walk stops there); tk = the token.
    // The entry layout (see the top of this file) makes this 13 vector instructions: a literal's entry IS its token, a
    // match's token is the two entries' high halves side by side plus the two extra-bit fields, and tb is a byte sum.
    PZG_FN void spec_finish(const Spec &t, uint32_t &tb, uint32_t &tk)
    {
        const uint32_t e = t.e, d = t.d;
        const uint32_t en = e >> 8, dn = d >> 8;              // [4:0] = code bits
        const uint32_t xl = ubfe(t.w_lo, en, e - en);         // length extra bits:   width = (n + extra) - n  (mod 32)
        const uint32_t xd = ubfe(t.w2, dn, d - dn);           // distance extra bits
        const uint32_t m = (uint32_t)((int32_t)e >> 31);      // all ones: a length entry, the distance entry counts
        // (round 4, by the measured issue costs -- profiles/r04_issue_ports.txt: a shift left, a three-operand add and anything with
        // a scalar operand are half-rate, a plain and / add of two registers is full-rate)
        const uint32_t tk_match = hi_halves(e, d) + shl16_add(xl, xd);  // TK_MATCH | (base len + xl) << 16 | (base dist + xd): no carries
        tb = byte0_sum(e, d & m);                             // byte sum (one SDWA add): either stop bit (0x80) makes it >= 128
        tk = bit_select(m, tk_match, e);
    }
    template <bool FX>
    PZG_FN void decode_at(uint32_t lo, uint32_t mid, uint32_t hi, uint32_t r, uint32_t &tb, uint32_t &tk)
    {
        Spec t;
        spec_bits<FX>(t, lo, mid, hi, r);
        if (!FX && use_sub) spec_sub(t);  // wave-uniform
        spec_dist<FX>(t);
        spec_finish(t, tb, tk);
    }
    // two independent decodes, stage by stage
    // SUB: 1 = the block's code has second-level tables, 0 = it has none, -1 = look at use_sub (the general paths)
    template <bool FX, int SUB = -1>
    PZG_FN void decode_pair(uint32_t lo0, uint32_t mid0, uint32_t hi0, uint32_t mid1, uint32_t hi1, uint32_t r, uint32_t &tb0,
                            uint32_t &tk0, uint32_t &tb1, uint32_t &tk1)
    {
        Spec a, b;
        spec_bits<FX>(a, lo0, mid0, hi0, r);
        spec_bits<FX>(b, hi0, mid1, hi1, r);
        if (!FX && (SUB < 0 ? use_sub != 0u : SUB != 0)) {  // wave-uniform
            uint32_t ea = spec_sub_load(a), eb = spec_sub_load(b);  // (both lookups in flight together)
#if PZG_DEVICE_PASS
            asm("" : "+v"(ea), "+v"(eb));
#endif
            spec_sub_pick(a, ea);
            spec_sub_pick(b, eb);
        }
        spec_dist<FX>(a);
        spec_dist<FX>(b);
        spec_finish(a, tb0, tk0);
        spec_finish(b, tb1, tk1);
    }

    // ---- phase B (scalar): follow the real chain through one 64-offset half --------------------------------
    // S collects the offsets visited from k on (the token starts).  Returns where the chain leaves the half MINUS 64:
    // below 64 it is the offset at which the next half is entered; 64 or more means the chain ran into a stopper (a
    // stopper's tb has bit 7 set; a token's is at most 48).
    // The only serial part of the decode, and the scalar unit is what the kernel runs out of first: two SALU
    // instructions, one v_readlane and one branch per token.  The offset is carried biased by -64 (mod 2^32), so the add
    // that advances it sets SCC exactly when the chain leaves the half -- no compare; s_bitset1 and v_readlane use the
    // low six bits of their index, which the bias leaves alone.
    PZG_FN uint32_t walk_half(const LaneVec<uint32_t> &TB, uint32_t k, uint64_t &S)
    {
#if PZG_DEVICE_PASS
        uint32_t kb = k | 0xffffffc0u, t;  // k - 64 for k < 64
        // (unrolled: a half holds six tokens on average, and a branch that falls through costs the wave no refetch)
        asm("1:\n\t"
            ".rept " PZG_STR(PZG_WALK_UNROLL) "\n\t"
            "s_bitset1_b64 %0, %1\n\t"
            "v_readlane_b32 %2, %3, %1\n\t"
            "s_add_u32 %1, %1, %2\n\t"
            "s_cbranch_scc1 2f\n\t"
            ".endr\n\t"
            "s_branch 1b\n\t"
            ".p2align " PZG_STR(PZG_WALK_EXIT_ALIGN) "\n"   // (experiment knob: padding here is never executed; measured neutral)
            "2:"
            : "+s"(S), "+s"(kb), "=&s"(t)
            : "v"(TB.v)
            : "scc");
        return kb;
#else
        do {
            S |= 1ull << k;
            k += lane_get(TB, k);
        } while (k < 64u);
        return k - 64u;
#endif
    }

    // ---- compaction: the token lanes of up to two halves go to the queue's tail, in order ---------------------
    // (one crossbar scatter per half; lanes that are not token starts send to lane 63, which is never a queue slot)
    PZG_FN void queue_append(const LaneVec<uint32_t> &TK0, uint64_t tokens0, uint32_t nt0, const LaneVec<uint32_t> &TK1,
                             uint64_t tokens1, uint32_t nt1)
    {
        LaneVec<uint32_t> DEST, R0, R1;
        PZG_LANES_BEGIN(k)
            PZG_LV(DEST, k) = mask_select(tokens0, k, mbcnt_slot4_k(tokens0, qn << 2, k), 63u << 2);
        PZG_LANES_END
        lanes_scatter4(R0, TK0, DEST);
        PZG_LANES_BEGIN(k)
            PZG_LV(DEST, k) = mask_select(tokens1, k, mbcnt_slot4_k(tokens1, (qn + nt0) << 2, k), 63u << 2);
        PZG_LANES_END
        lanes_scatter4(R1, TK1, DEST);
        // queue slots qn .. qn + nt0 - 1 take the first half's tokens, the next nt1 the second's (two scalar bit-field masks,
        // two selects; qn + nt0 + nt1 <= QCAP = 63)
        const uint64_t slots0 = bit_field_mask(nt0, qn), slots1 = bit_field_mask(nt1, qn + nt0);
        PZG_LANES_BEGIN(j)
            PZG_LV(QT, j) = mask_select2(slots0, slots1, j, PZG_LV(R0, j), PZG_LV(R1, j), PZG_LV(QT, j));
        PZG_LANES_END
        qn = (qn + nt0) + nt1;  // (associated as the callers' capacity tests are: one addition for both)
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
        prof[7] += nt0 + nt1;
#endif
    }
    // the offset of the token that no longer fits when only `room` of the tokens in `tokens` do
    PZG_FN uint32_t first_beyond(uint64_t tokens, uint32_t room)
    {
        LaneVec<bool> FIRST_OUT;
        PZG_LANES_BEGIN(k)
            PZG_LV(FIRST_OUT, k) = lane_bit(tokens, k) && mbcnt_k(tokens, k) == room;
        PZG_LANES_END
        return ctz64(lanes_ballot(FIRST_OUT));
    }

    // One 64-bit window (used where fewer than 320 stream bits are left for the 128-bit one).
    // Precondition: br.window_ok(): at least WINDOW_MIN_BITS real bits follow the cursor, so every token that
    // starts within the next 64 bits (at most 48 bits long) lies inside the stream; qn < QCAP.
    // Returns true when the token now at the cursor must go through token_step_checked().
    template <bool FX>
    PZG_FN bool window_append()
    {
        // lane k needs the three dwords that hold stream bits [k, k+96) from the cursor
        const uint32_t i0 = br.chunk0 + (br.rp >> 5), boff = br.rp & 31u;
        LaneVec<uint32_t> TB, TK;
        {
            const uint32_t B0 = br.dword(i0), B1 = br.dword(i0 + 1u), B2 = br.dword(i0 + 2u), B3 = br.dword(i0 + 3u),
                           B4 = br.dword(i0 + 4u);
            PZG_LANES_BEGIN(k)
                const uint32_t sel = (boff + k) >> 5, r = (boff + k) & 31u;
                const uint32_t lo = sel == 0u ? B0 : sel == 1u ? B1 : B2;
                const uint32_t mid = sel == 0u ? B1 : sel == 1u ? B2 : B3;
                const uint32_t hi = sel == 0u ? B2 : sel == 1u ? B3 : B4;
                decode_at<FX>(lo, mid, hi, r, PZG_LV(TB, k), PZG_LV(TK, k));
            PZG_LANES_END
        }
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
        prof[6] += 1;
#endif
        uint64_t S = 0;
        const uint32_t kend = walk_half(TB, 0u, S);  // (minus 64)
        bool stopper = kend >= 64u;
        uint32_t consumed = kend + 64u;
        uint64_t tokens = S;
        if (stopper) {  // the chain's last offset is the stopper: it stays at the cursor
            consumed = 63u - clz64(S);
            tokens = S & ~(1ull << consumed);
        }
        const uint32_t room = QCAP - 1u - qn;  // (one slot stays free: the token after a stopper is pushed by token_step_checked())
        uint32_t nt = popc64(tokens);
        if (nt > room) {  // (a window of very short codes) take what fits, the rest is decoded again
            consumed = first_beyond(tokens, room);
            tokens &= (1ull << consumed) - 1ull;
            nt = room;
            stopper = false;
        }
        queue_append(TK, tokens, nt, TK, 0ull, 0u);
        br.drop_short(consumed);
        return stopper;
    }

    // The hot loop's decode step, over 128 bits: lane k decodes at offsets k and k + 64.  The two decodes are
    // independent, so their LDS round trips overlap, and the per-window bookkeeping is paid once.
    // Precondition: br.window2_ok(); qn < QCAP.  Falls back to the first half alone when the queue
    // cannot take both.  Returns true when the token now at the cursor must go through token_step_checked().
    // the 128 speculative decodes of a window: lane k's tokens at bit offsets k (TB0, TK0) and k + 64 (TB1, TK1)
    // TAIL: the window may reach past the end of the stream.  A token that starts inside the stream and ends inside it was
    // decoded from stream bits alone (a prefix code is settled by its own bits, whatever follows them); any other is made
    // a stopper, so the walk ends in front of it and token_step_checked() finds what the reference finds there
    // (the end-of-block code as a rule; a truncated stream otherwise).
    template <bool FX, bool TAIL = false, int SUB = -1>
    PZG_FN void window2_decode(LaneVec<uint32_t> &TB0, LaneVec<uint32_t> &TK0, LaneVec<uint32_t> &TB1, LaneVec<uint32_t> &TK1)
    {
        PZG_MARK("w2.begin");
#if PZG_DEVICE_PASS
        {
            // lane k's first token starts at bit p = rp + k of the chunk: dword d = p >> 5 (the window's dwords all sit in `cur`:
            // chunks overlap, see BitReader), bit r = p & 31 of it.  The crossbar takes a byte address and ignores its low two bits
            // (lane = address[7:2]) and the funnel shift its amount's high bits, so p >> 3 and p themselves will do.
            static_assert((BitReader::STRIDE_BITS + 62u) / 32u + 4u <= 63u, "the last lane's fifth dword lies in `cur`");
            const uint32_t r = br.rp + lane_id(), a = r >> 3;
            uint32_t lo0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)a, (int)br.cur);
            uint32_t mid0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a + 4u), (int)br.cur);
            uint32_t hi0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a + 8u), (int)br.cur);
            uint32_t mid1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a + 12u), (int)br.cur);
            uint32_t hi1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a + 16u), (int)br.cur);
            decode_pair<FX, SUB>(lo0, mid0, hi0, mid1, hi1, r, TB0.v, TK0.v, TB1.v, TK1.v);
        }
#else
        {
            const uint32_t i0 = br.chunk0 + (br.rp >> 5), boff = br.rp & 31u;
            PZG_LANES_BEGIN(k)
                const uint32_t q = boff + k, d0 = i0 + (q >> 5), r = q & 31u;
                decode_pair<FX, SUB>(br.dword(d0), br.dword(d0 + 1u), br.dword(d0 + 2u), br.dword(d0 + 3u), br.dword(d0 + 4u), r,
                            PZG_LV(TB0, k), PZG_LV(TK0, k), PZG_LV(TB1, k), PZG_LV(TK1, k));
            PZG_LANES_END
        }
#endif
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
        prof[6] += 2;
#endif
        if (TAIL) {
            const int64_t av = br.avail();
            const uint32_t left = av < 0 ? 0u : av > 1024 ? 1024u : (uint32_t)av;  // stream bits from the cursor on
            PZG_LANES_BEGIN(k)
                PZG_LV(TB0, k) |= k + PZG_LV(TB0, k) > left ? ENT_STOP : 0u;
                PZG_LV(TB1, k) |= 64u + k + PZG_LV(TB1, k) > left ? ENT_STOP : 0u;
            PZG_LANES_END
        }
    }

    template <bool FX, bool TAIL = false>
    PZG_FN bool window_append2()
    {
        LaneVec<uint32_t> TB0, TK0, TB1, TK1;
        window2_decode<FX, TAIL>(TB0, TK0, TB1, TK1);
        // ---- the walk: first half, and unless it ended at a stopper, the second -------------------------
        // The common case is written straight through (no flags to merge): both halves walked, no stopper, the queue
        // takes every token.  Anything else goes through window2_rare().
        PZG_MARK("w2.walk");
        uint64_t S0 = 0, S1 = 0;
        const uint32_t k0 = walk_half(TB0, 0u, S0);  // where the chain enters the second half (64 or more: a stopper)
        if (__builtin_expect(k0 < 64u, 1)) {
            const uint32_t k1 = walk_half(TB1, k0, S1);
            if (__builtin_expect(k1 < 64u, 1)) {
                const uint32_t nt0 = popc64(S0), nt1 = popc64(S1);
                if (__builtin_expect(qn + nt0 + nt1 <= QCAP, 1)) {
        PZG_MARK("w2.append");
                    queue_append(TK0, S0, nt0, TK1, S1, nt1);
        PZG_MARK("w2.drop");
                    br.drop_short(k1 + 128u);
        PZG_MARK("w2.end");
                    return false;
                }
            }
            return window2_rare(TK0, TK1, S0, S1, k0, k1);
        }
        return window2_rare(TK0, TK1, S0, 0ull, k0, 0u);
    }

    // window_append2() when a half ended at a stopper or the queue cannot take every token
    PZG_FN bool window2_rare(const LaneVec<uint32_t> &TK0, const LaneVec<uint32_t> &TK1, uint64_t S0, uint64_t S1, uint32_t k0, uint32_t k1)
    {
        uint64_t tokens0 = S0, tokens1 = S1;
        uint32_t consumed = k1 + 128u;
        bool stopper = false;
        if (k0 >= 64u) {  // a stopper in the first half: the last offset visited; it stays at the cursor
            consumed = 63u - clz64(S0);
            tokens0 = S0 & ~(1ull << consumed);
            tokens1 = 0;
            stopper = true;
        } else if (k1 >= 64u) {  // ... in the second half
            const uint32_t p = 63u - clz64(S1);
            tokens1 = S1 & ~(1ull << p);
            consumed = p + 64u;
            stopper = true;
        }
        const uint32_t room = QCAP - 1u - qn;  // (one slot stays free: the token after a stopper is pushed by token_step_checked())
        uint32_t nt0 = popc64(tokens0), nt1 = popc64(tokens1);
        if (nt0 + nt1 > room) {  // not both halves: the first alone, the second is decoded again
            if (tokens1 != 0ull || !stopper) {  // (otherwise consumed / stopper already describe the first half alone)
                consumed = k0 + 64u;
                stopper = false;
            }
            tokens1 = 0;
            nt1 = 0;
            if (nt0 > room) {
                consumed = first_beyond(tokens0, room);
                tokens0 &= (1ull << consumed) - 1ull;
                nt0 = room;
                stopper = false;
            }
        }
        queue_append(TK0, tokens0, nt0, TK1, tokens1, nt1);
        br.drop_short(consumed);
        return stopper;
    }

    // Place and produce the bytes of the queue's leading tokens.  A segment is at most SEG = 128 output bytes, produced
    // in two passes of 64 (lane j: bytes j and 64 + j); the work per token is paid once for both.
    // Token lanes learn their output offset from a prefix sum of their lengths and announce themselves at that offset
    // (one crossbar scatter per pass); counting the announcements up to its own position tells every output lane which
    // token it belongs to; one ring gather (+ one far gather) and one ring store per pass follow.  A token ends the
    // segment (and heads the next one) if it does not fit the 128 bytes or is a match whose source is not complete
    // before the segment starts (dist < offset + len) or lies before the output (dist > produced + offset).  A match
    // that heads a segment and still does not fit overlaps its own output or is longer than the segment: copy_match()
    // takes it.  Precondition: qn != 0.
    static constexpr uint32_t SEG = 128u;

    // one pass of a segment, the gather: VAL = the byte of output offset o = 64 * pass + lane (from its literal token or
    // the near ring), FARM = the lanes whose source is older than the ring; DIST = their tokens' distances.
    PZG_FN uint64_t segment_gather(const LaneVec<uint32_t> &TOK, uint32_t o0, uint32_t run, uint32_t op32, LaneVec<uint32_t> &VAL,
                                   LaneVec<uint32_t> &DIST)
    {
        LaneVec<uint32_t> PJ;
        lanes_gather(PJ, QT, TOK);
        LaneVec<bool> INSIDE, MATCH, OLD;
        PZG_LANES_BEGIN(j)
            const uint32_t pj = PZG_LV(PJ, j), o = o0 + j, dist = pj & 0xffffu;
            const bool is_match = (int32_t)pj < 0;
            const uint8_t g = L.ring[(op32 + o - dist) & RMASK];
            PZG_LV(VAL, j) = is_match ? (uint32_t)g : ((pj >> 8) & 0xffu);
            PZG_LV(DIST, j) = dist;
            PZG_LV(INSIDE, j) = o < run;
            PZG_LV(MATCH, j) = is_match;
            PZG_LV(OLD, j) = dist - o > RING;
        PZG_LANES_END
        // (one ballot per compare and the masks combined by scalar instructions: a ballot of a compound predicate costs
        // two more vector instructions)
        return HYBRID ? (lanes_ballot(INSIDE) & lanes_ballot(MATCH) & lanes_ballot(OLD)) : 0ull;
    }
    // ... the stores: at once for the lanes that have their byte; the far lanes' bytes are requested and left pending.
    // The far request is ONE unconditional load per pass, straight-line (a load inside a branch makes the compiler merge
    // its result with the old register -- a copy that waits for the load on the spot): base + 32-bit lane offset, lanes
    // with no far source read FAR_IDLE (see set_far_base: a byte that is there, in a line that is final).
    PZG_FN void segment_store(uint32_t o0, uint32_t run, uint32_t op32, const LaneVec<uint32_t> &VAL, const LaneVec<uint32_t> &DIST,
                              uint64_t farm, uint32_t fdelta, LaneVec<uint8_t> &pendF)
    {
        PZG_LANES_BEGIN(j)
            const uint32_t o = o0 + j;
            ring_store((o < run) & !lane_bit(farm, j), (op32 + o) & RMASK, (uint8_t)PZG_LV(VAL, j), j);
        PZG_LANES_END
        if (HYBRID) {  // sources older than the ring: the stream's own flushed output (fdelta = op - flushed)
            if (RES_HIST) {  // the decoder's history, by position modulo its size
                PZG_LANES_BEGIN(j)
                    const uint32_t hp = lane_bit(farm, j) ? (op32 + (o0 + j) - PZG_LV(DIST, j)) & HIST_MASK : 0u;
#if PZG_DEVICE_PASS
                    PZG_LV(pendF, j) = __builtin_nontemporal_load(hist + hp);
#else
                    PZG_LV(pendF, j) = hist[hp];
#endif
                PZG_LANES_END
                return;
            }
            PZG_LANES_BEGIN(j)
                const uint32_t off = lane_bit(farm, j) ? 32768u + fdelta + (o0 + j) - PZG_LV(DIST, j) : FAR_IDLE;
#if PZG_DEVICE_PASS
                PZG_LV(pendF, j) = far_base[off];
#else
                PZG_LV(pendF, j) = off != FAR_IDLE ? far_base[off] : (uint8_t)0;
#endif
            PZG_LANES_END
        }
    }


    // Resumable instance: the reference publishes its output in 32 KiB chunks -- moveWindow (Monad.hs:338-347) runs after
    // every match and at every block end and hands out ONE chunk when the window holds 64 KiB or more
    // (OutputWindow.hs:45-54).  `ow` follows the window fill and `chunks` the count, token by token where it matters, so
    // that the host publishes exactly the chunks the reference has published when the input runs out.
    PZG_FN void account_tokens(uint32_t run, uint32_t v, bool single_match)
    {
        if (single_match) {
            ow += run;
            move_window_check();
            return;
        }
        if (ow + run < 65536u) {  // no check in this segment can fire
            ow += run;
            return;
        }
        for (uint32_t t = 0; t < v; ++t) {  // (the queue has not moved up yet when emit_segment calls this)
            const uint32_t tk = lane_get(QT, t);
            ow += (tk >> 16) & 511u;
            if ((int32_t)tk < 0) move_window_check();
        }
    }
    PZG_FN void move_window_check()
    {
        if (ow >= 65536u) {
            ow -= 32768u;
            chunks += 1u;
        }
    }
    PZG_FN int emit_segment() { return emit_body<false>(); }

    // FAST (hot_loop): returns EMIT_BAIL, with nothing changed that emit_segment() would not redo, where the general
    // code has lane-dependent branches -- a flush is due, or the queue's head is a match for copy_match().
    static constexpr int EMIT_BAIL = -1;
    template <bool FAST>
    PZG_FN int emit_body()
    {
        PZG_MARK("e.begin");
        PZG_T0(t_a);
        // the previous segment's bytes must all be in the ring from here on
        if (FAST) {
            // (tried with the strips: waiting for the last segment's far bytes only in front of this segment's first ring read, and
            // a far fence only after a flush -- nothing for the strips, and the windows' loop lost 20 % to the changed code)
            pending_stores();
            if (__builtin_expect((uint32_t)(op - flushed) >= FLUSH_AT, 0)) {  // whole KiB only (the general flush goes up to op & ~15)
                const uint64_t to = flushed + ((uint32_t)(op - flushed) & ~1023u);
                if (!out_aligned() || to > cap) return EMIT_BAIL;
                flush_span<true>(to);
            }
        } else {
            complete_pending();
        }
        PZG_ACCW(8, t_a);
        if (RES && op + 512u > cap) return FAST ? EMIT_BAIL : (int)ST_OUT_FULL;  // (resumable: never produce past this call's output room)
        PZG_MARK("e.scan");
        PZG_T0(t_b);
        // bytes of history a distance may reach back over; dist <= 32768, so a clamp is enough (scalar shift + test)
        const uint32_t op32 = (uint32_t)op;
        uint32_t op_hi = (uint32_t)(op >> 32);
#if PZG_DEVICE_PASS
        asm("" : "+s"(op_hi));  // (opaque: `op >> 20` as written becomes a 64-bit VECTOR compare against a constant in a register pair)
#endif
        const uint32_t hist = (op_hi | (op32 >> 20)) ? 0x100000u : op32 + (RING_BITS == 15 ? hist_extra : 0u);
        LaneVec<uint32_t> INCL, START;
        LaneVec<bool> BIG, MATCH, SRC_IN, SRC_OUT;
        const uint64_t waiting = bit_field_mask(qn, 0u);  // lanes 0 .. qn - 1 (qn <= 63)
        PZG_LANES_BEGIN(t)
            PZG_LV(INCL, t) = t < qn ? ((PZG_LV(QT, t) >> 16) & 511u) : 0u;
        PZG_LANES_END
        lanes_iscan_add(INCL);
        PZG_LANES_BEGIN(t)
            const uint32_t tk = PZG_LV(QT, t), lout = (tk >> 16) & 511u, dist = tk & 0xffffu;
            const uint32_t endb = PZG_LV(INCL, t), start = endb - lout;
            PZG_LV(START, t) = start;
            PZG_LV(BIG, t) = endb > SEG;                 // does not fit the segment
            PZG_LV(MATCH, t) = (int32_t)tk < 0;
            PZG_LV(SRC_IN, t) = dist < endb;              // a match whose source is not complete before the segment starts
            PZG_LV(SRC_OUT, t) = dist > hist + start;     // ... or lies before the output (an error: found when it heads a segment)
        PZG_LANES_END
        // (four compares, the rest on the masks: scalar instructions instead of select / or / compare chains per lane)
        const uint64_t stopmask =
            waiting & (lanes_ballot(BIG) | (lanes_ballot(MATCH) & (lanes_ballot(SRC_IN) | lanes_ballot(SRC_OUT))));
        uint32_t v = stopmask ? ctz64(stopmask) : qn;  // tokens of this segment
#if defined(PZG_STATS) && !PZG_DEVICE_PASS
        PZG_STAT(2, 1);  // segments (FAST and general)
        PZG_STAT(3, FAST ? 1 : 0);
        if (stopmask) {
            const uint64_t first = stopmask & (0ull - stopmask);
            PZG_STAT(4, (first & lanes_ballot(BIG)) ? 1 : 0);  // ended by the 128-byte limit
            PZG_STAT(5, (!(first & lanes_ballot(BIG)) && (first & lanes_ballot(SRC_IN))) ? 1 : 0);  // ... by a source inside the segment
        } else {
            PZG_STAT(6, 1);  // the queue ran out
        }
        PZG_STAT(7, qn);
#endif
        PZG_MARK("e.v");
        PZG_ACCW(9, t_b);
        if (__builtin_expect(v == 0u, 0)) {
            PZG_STAT(11, 1);  // a head token for copy_match (or a bail-out of the fast body)
            if (FAST) return EMIT_BAIL;
            const uint32_t tk = lane_get(QT, 0u), dist = tk & 0xffffu, len = (tk >> 16) & 511u;
            if ((uint64_t)dist > op + (RING_BITS == 15 ? hist_extra : 0u)) return fail(ST_BAD_DISTANCE, dist, (uint32_t)op);
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
            prof[14] += 1;
#endif
            copy_match(dist, len);
            maybe_flush();
            v = 1u;
            if (RES) account_tokens(len, 1u, true);
        } else {
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
            prof[13] += 1;
#endif
            PZG_T0(t_d);
            const uint32_t run = lane_get(INCL, v - 1u);
            PZG_STAT(8, run);   // bytes of non-degenerate segments
            PZG_STAT(9, v);     // their tokens
            PZG_STAT(10, run > 64u ? 1 : 0);
            if (run == v) {
                // one byte per token: nothing but literals (a match is three bytes or more).  Byte j IS token j's byte:
                // no announcements, no gathers -- literal-heavy data (little or no redundancy) spends its time here.
                PZG_LANES_BEGIN(j)
                    ring_store(j < v, ((op32 + j)) & RMASK, (uint8_t)(PZG_LV(QT, j) >> 8), j);
                PZG_LANES_END
                op += run;
                if (RES) ow += run;  // (no match: no moveWindow check)
                PZG_ACC(11, t_d);
            } else {
            // Which token does output byte o belong to?  Token t announces itself at lane START[t] of its pass (one
            // crossbar scatter); the tokens sit in the queue in output order, so byte o belongs to token
            // (number of announcements at offsets <= o) - 1.  Lanes that send nothing real repeat an announcement that
            // is made anyway -- token 0's at lane 0 in the first pass, the last token's in the second -- so colliding
            // writes all carry the same value.
            LaneVec<uint32_t> ONE, DEST, MARK, TOK, VAL0, VAL1, DIST0, DIST1;
        PZG_MARK("e.general");
            PZG_LANES_BEGIN(t)
                PZG_LV(DEST, t) = ((t < v) & (PZG_LV(START, t) < 64u)) ? PZG_LV(START, t) : 0u;
                PZG_LV(ONE, t) = 1u;
            PZG_LANES_END
            lanes_scatter(MARK, ONE, DEST);
            const uint64_t starts0 = lanes_ballot(MARK);  // (bit 0 is always set)
            PZG_LANES_BEGIN(j)
                PZG_LV(TOK, j) = mbcnt_k(starts0 >> 1, j);  // announcements at offsets 1 .. j
            PZG_LANES_END
        PZG_MARK("e.gather0");
            uint64_t farm0 = segment_gather(TOK, 0u, run, op32, VAL0, DIST0);
        PZG_MARK("e.pass1");
            uint64_t farm1 = 0ull;
            if (run > 64u) {
                const uint32_t start_last = lane_get(START, v - 1u);
                uint64_t starts1 = 0ull;
                if (start_last >= 64u) {  // (otherwise the last token covers the whole second pass)
                    PZG_LANES_BEGIN(t)
                        PZG_LV(DEST, t) = (((t < v) & (PZG_LV(START, t) >= 64u)) ? PZG_LV(START, t) : start_last) - 64u;
                    PZG_LANES_END
                    lanes_scatter(MARK, ONE, DEST);
                    starts1 = lanes_ballot(MARK);
                }
                // token of offset 64: the last one announced in the first pass, or the one announced at 64 itself
                const uint32_t base1 = popc64(starts0) - 1u + ((uint32_t)starts1 & 1u);
                PZG_LANES_BEGIN(j)
                    PZG_LV(TOK, j) = base1 + mbcnt_k(starts1 >> 1, j);
                PZG_LANES_END
                farm1 = segment_gather(TOK, 64u, run, op32, VAL1, DIST1);
            }
        PZG_MARK("e.store");
            // both gathers precede the stores: the first pass' bytes replace ring bytes the second may still read
            farm0 &= far_okmask;
            farm1 &= far_okmask;
            const uint32_t fdelta = (uint32_t)(op - flushed);
            if (HYBRID && (farm0 | farm1) != 0ull) far_fence();
            segment_store(0u, run, op32, VAL0, DIST0, farm0, fdelta, pendF0);
            if (run > 64u) segment_store(64u, run, op32, VAL1, DIST1, farm1, fdelta, pendF1);
        PZG_MARK("e.stored");
            pend_m0 = farm0;
            pend_m1 = farm1;
            pend_pos = op32;
            op += run;
            if (RES) account_tokens(run, v, false);
            PZG_ACC(11, t_d);
            }
        }
        PZG_MARK("e.shift");
        // the queue moves up by v tokens
        LaneVec<uint32_t> SRC;
        PZG_LANES_BEGIN(j)
            PZG_LV(SRC, j) = j + v;
        PZG_LANES_END
        lanes_gather(QT, QT, SRC);
        PZG_MARK("e.end");
        qn -= v;
        return ST_OK;
    }

    // Fill the queue: 128-bit windows while at least 320 stream bits are ahead, then 64-bit ones.  Returns true when
    // the token at the cursor is one for token_step_checked() (a window said so, or too few bits are left for one).
    template <bool FX>
    PZG_FN bool fill_queue()
    {
        // the hot loop proper: nothing but 128-bit windows
        while (__builtin_expect(br.window2_ok(), 1)) {
            if (__builtin_expect(window_append2<FX>(), 0)) return true;
            if (qn >= QHIGH) return false;
        }
        // the last 320 bits of the stream
        if (RES) {  // (the resumable instance stops short of the end of its input: more may follow)
            do {
                if (!br.window_ok() || window_append<FX>()) return true;
            } while (qn < QHIGH);
            return false;
        }
        do {  // windows that may reach past the end: they stop in front of the first token that does
            if (window_append2<FX, true>()) return true;
        } while (qn < QHIGH);
        return false;
    }

    // The hot loop: 128-bit windows while the queue is short of QHIGH tokens, segments while it is not, for as long as
    // neither needs code with a lane-dependent branch.  The compiler rebuilds the control flow of every region that holds
    // one such branch with flags and flag tests (its structurizer), including all the wave-uniform branches around it:
    // ~35 scalar instructions per window in the token loop as a whole.  This loop holds none, has ONE exit, and the reason
    // for leaving passes through an opaque statement (nothing can be threaded from inside the loop to the handlers), so it
    // stays a region of plain scalar branches.  HL_WINDOW: the window decoded last (TK0, TK1, S0, S1, k0, k1) is one for
    // window2_rare(); HL_GENERAL: the general token loop has to take a step (stream tail, flush, copy_match).
    // Round 3 tried to take scalar work out of this loop (one compare per window for the cursor, one for the queue, use_sub a
    // template argument: -2 % scalar, -11 % branch instructions by the counters) and measured no gain (DESIGN.md 8,
    // profiles/r03_costmodel.txt); two things learnt on the way, for whoever edits it next: (1) every way out must be a
    // `break` to the ONE block behind the loop with its reason in `why` -- give the exits blocks of their own, or put an inner
    // loop with two ways out inside, and the compiler funnels them through a dispatch variable that EVERY iteration sets and
    // tests; (2) a uniform value the register allocator parks in a vector register (the scalar file is full here) makes its
    // compare a vector compare and the branch, and with it the whole region, divergent for the structurizer: pass such
    // operands through uni() at the compare, as window2_ok() does.
    // Round 4: the loop exists once per kind of block -- fixed code, dynamic code with and without second-level tables
    // (SUB) -- so that the windows test nothing about the block: two scalar instructions per window less.
    enum : uint32_t { HL_GENERAL = 1, HL_WINDOW = 2 };
    template <bool FX, int SUB>
    PZG_FN uint32_t hot_loop(LaneVec<uint32_t> &TK0, LaneVec<uint32_t> &TK1, uint64_t &S0, uint64_t &S1, uint32_t &k0, uint32_t &k1)
    {
        uint32_t why;
#if PZG_DEVICE_PASS && PZG_HOT_ALIGN
        // the loop starts on a fixed boundary, so that where its blocks fall within the instruction-fetch blocks does not
        // move with every edit of the code in front of it (measured neutral at 0 / 64 / 128 / 256 bytes on this build)
        asm volatile(".p2align " PZG_STR(PZG_HOT_ALIGN));
#endif
        for (;;) {
            if (qn < QHIGH) {
                if (__builtin_expect(!br.window2_ok(), 0)) {
                    why = HL_GENERAL;
                    break;
                }
                PZG_T0(thw);
                LaneVec<uint32_t> TB0, TB1;
                window2_decode<FX, false, SUB>(TB0, TK0, TB1, TK1);
                S0 = 0;
                S1 = 0;
                k1 = 0;
                PZG_STAT(0, 1);  // windows decoded by the hot loop
                k0 = walk_half(TB0, 0u, S0);
                why = HL_WINDOW;
                if (__builtin_expect(k0 >= 64u, 0)) break;
                k1 = walk_half(TB1, k0, S1);
                if (__builtin_expect(k1 >= 64u, 0)) break;
                const uint32_t nt0 = popc64(S0), nt1 = popc64(S1);
                if (__builtin_expect((qn + nt0) + nt1 > QCAP, 0)) break;
                PZG_STAT(1, nt0 + nt1);  // tokens queued by clean windows
                queue_append(TK0, S0, nt0, TK1, S1, nt1);
                br.drop_short(k1 + 128u);
                PZG_HOT_ACC(8, thw);
                continue;
            }
            PZG_T0(the);
            if (emit_body<true>() != ST_OK) {
                why = HL_GENERAL;
                break;
            }
            PZG_HOT_ACC(9, the);
        }
#if PZG_DEVICE_PASS
        asm volatile("" : "+s"(why));
#endif
        return why;
    }


    // ---- strips (round 4; round 5: sequences): a long run of input decoded by 64 lanes side by side -------------------
    // The windows pay ~140 instructions per 128 input bits -- 128 speculative lane-decodes and a serial walk for ~12 real
    // tokens.  A strip pays ~70 instructions for 64 REAL tokens: the input in front of the cursor is cut into 64 strips of C
    // bits, and lane k decodes strip k token by token (the same LDS lookups and 13 vector instructions as a window's lane),
    // reading its input through a 192-bit buffer of its own that moves on 64 bits at a time.  What a lane cannot know is where
    // its first token starts.  Phase A: it starts STRIP_BACK bits in front of its strip at an arbitrary bit and decodes up to
    // the strip -- DEFLATE's codes re-synchronise: by then the lane is on the stream's real chain of tokens with probability
    // ~0.996 for text (measured: half of all wrong starts are back on the chain after 92 bits, 99 % after 630).  Phase B:
    // every lane decodes its strip from there; where lane k - 1's chain leaves its strip must be where lane k started -- lanes
    // for which that is not so decode again from the right place (and then perhaps the next lane...), at most STRIP_ROUNDS
    // times, after which the span simply ends in front of the first lane that is still wrong.  It also ends at the first lane
    // that met a stopper (end of block, a code the tables do not resolve, an error: token_step_checked()'s business, as with
    // the windows).  Nothing about the result depends on the guesses: a wrong guess costs a round, never
    // a token.
    // Round 5: what phase B writes to the wave's scratch is no longer one dword per token but the stream as SEQUENCES, the way
    // an LZ77 copier wants it (Deflate.hs:106-120 runInflate alternates exactly these two actions): a run of literal bytes
    // followed by one match.  A lane's region holds a literal area (one byte per literal, stored 16 at a time) and a record
    // area (one dword per sequence: literal-run length, match length, distance; stored SEQ_G at a time).  seq_group() then
    // gives every LANE one whole SEQUENCE -- round 4's segments gave every lane one output BYTE and paid a prefix scan, two
    // crossbar scatters and two byte gathers per ~100 bytes: 2.3 wave-instructions per output byte, 77 % of a stream's time.
    // A group of up to 64 sequences (~500 bytes of text) costs one prefix sum, and the bytes move as unaligned dwords.
    // (round 5: the resumable instance takes them too -- a span is decoded and emitted inside ONE call: it is cut behind the last lane
    // whose output still fits the call's room, its far matches read the decoder's history, and the reference's chunk count is kept)
    static constexpr bool STRIPS = true;
#ifndef PZG_STRIP_TMAX
#define PZG_STRIP_TMAX 192  // (round 4, measured on text / html / mixed / literal-heavy / config 3: 128: 249 / 237 / 260 / 128 / 141 GiB/s; 160: 260 / 242 /
                            // 269 / 123 / 139; 192: 263 / 241 / 272 / 134 / 141; 224: 261 / 243 / 277 / 127 / 138; 256: 260 / 247 / 277 / 120 / 141)
#endif
#ifndef PZG_STRIP_BACK
#define PZG_STRIP_BACK 1024  // (round 5, once a 32 KiB stream is ONE span: a repair re-decodes a strip twice as long as before -- 768: 300.4, 1024: 305.1 GiB/s)
#endif
    static constexpr uint32_t STRIP_TMAX = PZG_STRIP_TMAX;      // tokens one lane may decode per span (a multiple of 16)
    // (a strip of C <= STRIP_TMAX x the shortest code bits holds at most STRIP_TMAX tokens, so at most as many records and literal
    // bytes: a region cannot overflow, and a run of literals inside one strip cannot outgrow the record's eight-bit count)
    static_assert(STRIP_TMAX % 16u == 0u && STRIP_TMAX <= 1008u, "region geometry");
    static constexpr bool SEQ_LROVF = STRIP_TMAX > 240u;  // a record counts up to 255 literals: a longer run is cut into records without a match
#ifndef PZG_SEQ_GROUP
#define PZG_SEQ_GROUP 8
#endif
    // A lane's region of the scratch, byte offsets: 16 bytes of slack | the literal area (STRIP_TMAX bytes from REG_LITA on) |
    // SEQ_G records of slack | the record area (STRIP_TMAX dwords from REG_RECA on).  The slack in front of an area takes the
    // lane's last, partial group, which is stored as the lane's LAST 16 literals / SEQ_G records wherever they end.
    static constexpr uint32_t SEQ_G = PZG_SEQ_GROUP;           // records per store (a power of two)
    static constexpr uint32_t REG_LITA = 32u;
    static constexpr uint32_t REG_RECA = (REG_LITA + STRIP_TMAX + 4u * SEQ_G + 31u) & ~31u;
#ifndef PZG_REG_PAD
#define PZG_REG_PAD 0
#endif
    static constexpr uint32_t REG_BYTES = ((REG_RECA + 4u * STRIP_TMAX + 63u) & ~63u) + PZG_REG_PAD;
    // (the regions | 128 dwords that a refill may read past their end | the wave's profile, see strip_profile_*)
    static constexpr uint32_t PROF_OFF = 64u * (REG_BYTES / 4u) + 128u;
    static constexpr uint32_t STRIP_WORDS = PROF_OFF + 80u;  // dwords of scratch per wave
    static constexpr uint32_t PROF_MAGIC = 0x51DF0A7Eu;
#ifndef PZG_STRIP_PROFILE
#define PZG_STRIP_PROFILE 1
#endif
    static constexpr bool STRIP_PROFILE = PZG_STRIP_PROFILE != 0 && !RES;
    // (the profile's code runs once per stream: laid out as the unlikely side of its branches it stays out of the way of everything
    // that runs per token -- see DESIGN.md 4.1 for what its mere presence cost the windows' loop before)
#ifndef PZG_PROF_LIKELY
#define PZG_PROF_LIKELY(x) __builtin_expect(!!(x), 0)
#endif
    static constexpr uint32_t PROF_TGT = STRIP_TMAX * 3u / 4u;  // tokens per lane that a span laid out by the profile aims at
#ifndef PZG_PROF_CMIN
#define PZG_PROF_CMIN 512
#endif
    static constexpr uint32_t PROF_CMIN = PZG_PROF_CMIN;        // ... and only streams with so many bits per lane have one
    static constexpr uint32_t PROF_HEAD = 2048u;                // a span that starts within so many bits of the stream's start is "the first"
    static constexpr uint32_t STRIP_BACK = PZG_STRIP_BACK;      // the run-up of phase A, in bits: where a wave starts, and the most it uses
    // Round 5: the run-up ADAPTS.  How fast a wrong start falls back onto the real chain of tokens depends on the code: measured
    // with a fixed run-up of 256 / 384 / 512 / 768 bits, text 257 / 261 / 268 / 276 GiB/s and config 3 119 / 136 / 144 / 148 (a lane
    // that starts wrong costs a whole round of phase B) -- but literal-heavy data 139 / 136 / 133 / 128: its short literal codes
    // re-synchronise within a few dozen bits and the long run-up is wasted work.  So a span that needed no repair shortens the
    // wave's run-up by STRIP_BACK_DOWN bits, one that did lengthens it by STRIP_BACK_UP: it settles where about one span in
    // (1 + UP / DOWN) needs a repair.  The value lives in the wave's LDS across streams (a batch is mostly one kind of data).
    // Nothing about the result depends on it.
#ifndef PZG_STRIP_BACK_MIN
#define PZG_STRIP_BACK_MIN 256
#endif
    static constexpr uint32_t STRIP_BACK_MIN = PZG_STRIP_BACK_MIN < PZG_STRIP_BACK ? PZG_STRIP_BACK_MIN : PZG_STRIP_BACK;
    static constexpr uint32_t STRIP_BACK_DOWN = 32u, STRIP_BACK_UP = 256u;
#ifndef PZG_STRIP_CMIN
#define PZG_STRIP_CMIN 64  // (256 until round 6: set when a span's tokens were emitted by segments; with the groups, strips of a few tokens pay)
#endif
    static constexpr uint32_t STRIP_CMIN = PZG_STRIP_CMIN;      // shorter strips are not worth a span
#ifndef PZG_STRIP_BACK_PER_C
#define PZG_STRIP_BACK_PER_C 2
#endif
    static constexpr uint32_t STRIP_BACK_PER_C = PZG_STRIP_BACK_PER_C;  // the run-up of short strips, in strips
#ifndef PZG_STRIP_ROUNDS
#define PZG_STRIP_ROUNDS 6
#endif
    static constexpr uint32_t STRIP_ROUNDS = PZG_STRIP_ROUNDS;
    static constexpr int STRIP_NA = -2;
    PZG_FN static constexpr uint32_t reg_rec(uint32_t k) { return k * (REG_BYTES / 4u) + REG_RECA / 4u; }  // dword index of region k's first record
    PZG_FN static constexpr uint32_t reg_lit(uint32_t k) { return k * REG_BYTES + REG_LITA; }             // byte offset of its first literal

    // A sequence record:  [14:0] distance - 1   [22:15] match length - 3   [30:23] literals in front of the match (0..255)
    //                     [31] no match follows (the lane's strip ended in literals)
    static constexpr uint32_t SEQ_NOMATCH = 0x80000000u;
    PZG_FN static uint32_t seq_nl(uint32_t rec) { return (rec >> 23) & 255u; }
    PZG_FN static uint32_t seq_len(uint32_t rec) { return (int32_t)rec < 0 ? 0u : ((rec >> 15) & 255u) + 3u; }
    PZG_FN static uint32_t seq_dist(uint32_t rec) { return (rec & 0x7fffu) + 1u; }

    // the wave's own stores of a moment ago, read back by OTHER lanes: device-scope loads (not served from a stale L1 line)
    PZG_FN uint32_t strip_load(uint32_t i) const
    {
#if PZG_DEVICE_PASS
        return __hip_atomic_load(strip + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        return strip[i];
#endif
    }
    // four / one literal bytes at byte offset `off` of the scratch (any alignment: the hardware takes unaligned dword loads)
    PZG_FN uint32_t lit_load32(uint32_t off) const
    {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(strip) + off;
#if PZG_DEVICE_PASS
        return __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        uint32_t v;
        __builtin_memcpy(&v, p, 4);
        return v;
#endif
    }
    PZG_FN uint8_t lit_load8(uint32_t off) const
    {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(strip) + off;
#if PZG_DEVICE_PASS
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        return *p;
#endif
    }
    PZG_FN void strip_fence() const
    {
#if PZG_DEVICE_PASS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    // The regions that hold records, in order: CRIDX[t] = the lane of the t-th one, CSTA[t] = the span's records in front of it
    // (lanes behind the last one: never reached).  Built once per span from the lanes' record counts.
    PZG_FN void seq_index(const LaneVec<uint32_t> &NR, uint32_t last)
    {
        LaneVec<uint32_t> PRE, EXCL, DEST, KIDX, A, B;
        LaneVec<bool> NE;
        PZG_LANES_BEGIN(k)
            const uint32_t n = k <= last ? PZG_LV(NR, k) : 0u;
            PZG_LV(PRE, k) = n;
            PZG_LV(NE, k) = n != 0u;
        PZG_LANES_END
        const uint64_t ne = lanes_ballot(NE);
        s_cnt = popc64(ne);
        lanes_iscan_add(PRE);
        s_total = lane_get(PRE, 63u);
        PZG_LANES_BEGIN(k)
            const uint32_t n = k <= last ? PZG_LV(NR, k) : 0u;
            PZG_LV(EXCL, k) = PZG_LV(PRE, k) - n;
            PZG_LV(KIDX, k) = k;
            PZG_LV(DEST, k) = lane_bit(ne, k) ? mbcnt_k(ne, k) : 63u;  // (lanes without records send to lane 63: a slot only when there are none)
        PZG_LANES_END
        lanes_scatter(A, EXCL, DEST);
        lanes_scatter(B, KIDX, DEST);
        PZG_LANES_BEGIN(t)
            PZG_LV(CSTA, t) = t < s_cnt ? PZG_LV(A, t) : 0xffffffffu;
            PZG_LV(CRIDX, t) = t < s_cnt ? PZG_LV(B, t) : 0u;
        PZG_LANES_END
    }
    // The next group's records (QTN, s_qn) = the span's records from number s_rd on, from however many regions they lie in:
    // the regions that start inside the group announce themselves at the lane they start at (one crossbar scatter), a count of
    // the announcements up to its own position tells every lane its region.  True: record s_rd is the first of its region.
    PZG_FN bool seq_refill()
    {
        const uint32_t left = s_total - s_rd, n = left < 64u ? left : 64u;
        s_qn = n;
        if (n == 0u) return false;
        LaneVec<bool> BELOW;
        PZG_LANES_BEGIN(t)
            PZG_LV(BELOW, t) = PZG_LV(CSTA, t) <= s_rd;
        PZG_LANES_END
        const uint32_t t0 = popc64(lanes_ballot(BELOW)) - 1u;  // (the first region starts at 0: the count is 1 or more)
        LaneVec<uint32_t> DEST, ONE, MARK, TJ, RJ, SJ;
        PZG_LANES_BEGIN(t)
            const uint32_t b = PZG_LV(CSTA, t) - s_rd;  // where region t starts, relative to the group
            PZG_LV(DEST, t) = b - 1u < 63u ? b : 0u;    // (inside the group, not at its first lane: 1..63)
            PZG_LV(ONE, t) = 1u;
        PZG_LANES_END
        lanes_scatter(MARK, ONE, DEST);
        const uint64_t marks = lanes_ballot(MARK) >> 1;  // (lane 0 collects what is not sent anywhere)
        PZG_LANES_BEGIN(j)
            PZG_LV(TJ, j) = t0 + mbcnt_k(marks, j);
        PZG_LANES_END
        lanes_gather(RJ, CRIDX, TJ);
        lanes_gather(SJ, CSTA, TJ);
        PZG_LANES_BEGIN(j)
            const uint32_t rj = PZG_LV(RJ, j), rel = PZG_LV(SJ, j) - s_rd;  // (signed: the group's first region starts at or in front of it)
            const uint32_t fl = (int32_t)rel > 0 ? rel : 0u;
            PZG_LV(QTN, j) = strip_load(j < n ? reg_rec(rj) + (j - rel) : reg_rec(0u));
            PZG_LV(QINFO, j) = rj | (fl << 8) | ((PZG_LV(TJ, j) == t0 ? 1u : 0u) << 16);
        PZG_LANES_END
        return lane_get(CSTA, t0) == s_rd;
    }
    // the distance code's second level, as spec_sub() for the literal/length code (a distance base can have bit 30 set,
    // so K_SUB is recognised by its stop bit and kind)
    PZG_FN void spec_dsub(Spec &t)
    {
        const bool is_sub = (t.d & 0x780u) == 0x780u;
        const uint32_t off = ((t.d >> 14) & 0x3fcu) + (ubfe(t.w2, DIST_BITS, t.d) << 2);
#if PZG_DEVICE_PASS
        uint32_t d2 = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(L.sub) + off);
        asm("" : "+v"(d2));
#else
        const uint32_t d2 = is_sub ? L.sub[off >> 2] : 0u;
#endif
        t.d = is_sub ? d2 : t.d;
    }
    // A lane's reader: W0, W1 = the pairs of dwords (64 bits each) that hold the lane's position, r = that position relative to W0's bit 0
    // (below 64 when a token is decoded: a token is at most 48 bits, r + 48 <= 112, so W0 and W1 hold it), T = the pair behind W1, nx = T's
    // dword index, TN = the pair that is landing.  (Pairs past `maxdw` -- the stream's last dwords -- repeat that pair: no token of a span
    // reaches there.)  At the top of a step a lane that has moved past W0 takes W1 for W0 and T for W1 and asks for the pair behind
    // them, which becomes T at the top of the next step; the fetch is straight-line code (round 4, measured: written as a load inside a
    // lane-dependent branch it goes to a temporary that is copied into place at once -- a wait for HBM in every step): every lane loads
    // every step, the lanes that did not move the span's first pair.
    // Rounds 4-5 kept two pairs in flight in two slots, with two landing registers that alternated between the steps, so that a pair was
    // asked for two steps before its use: 16 registers and 24 instructions a step.  Measured at the end of round 5: the second step of
    // distance buys nothing (this reader: text 307.3, the same with two alternating landing registers and a lane that waits a step out
    // when it moves twice in a row: 305.2; config 3 158.5 / 154.6) -- 11 registers, 13 instructions: config 3 148 -> 158 GiB/s, the rest
    // +0.5 %.  What does NOT work is every lane asking for its own pair again every step (10 instructions): 64 lines through the first
    // cache per step, text 304 -> 182.
    struct StripReader {
        LaneVec<uint64_t> W0, W1, T, TN;  // ... TN: the pair that is landing; it becomes T at the top of the next step if PEND says so
        LaneVec<uint32_t> R, NX;
        LaneVec<bool> PEND;
    };
    PZG_FN static uint64_t strip_pair(const uint32_t *sp, uint32_t i)
    {
        const uint32_t *q = (const uint32_t *)(const void *)((const uint8_t *)(const void *)sp + (i << 2));
        return (uint64_t)q[0] | ((uint64_t)q[1] << 32);
    }
    // position a lane's reader at bit p of the span
    PZG_FN static void strip_open(const uint32_t *sp, uint32_t maxdw, uint32_t p, uint64_t &w0, uint64_t &w1, uint64_t &t, uint32_t &r, uint32_t &nx, bool &pend)
    {
        pend = false;
        const uint32_t g2 = (p >> 6) << 1;
        w0 = strip_pair(sp, g2 < maxdw ? g2 : maxdw);
        w1 = strip_pair(sp, g2 + 2u < maxdw ? g2 + 2u : maxdw);
        t = strip_pair(sp, g2 + 4u < maxdw ? g2 + 4u : maxdw);
        r = p & 63u;
        nx = g2 + 4u;
    }
    // the top of a step: 64 bits on if the position says so, and the pair behind W1 asked for (again)
    PZG_FN static void strip_top(const uint32_t *sp, uint32_t maxdw, uint64_t &w0, uint64_t &w1, uint64_t &t, uint64_t &tn, uint32_t &r, uint32_t &nx, bool &pend)
    {
        t = pend ? tn : t;
        const bool sh = r >= 64u;
        w0 = sh ? w1 : w0;
        w1 = sh ? t : w1;
        r &= 63u;
        nx += sh ? 2u : 0u;
        // (the lanes that did not move ask for the span's first pair: one cache line for all of them -- asking for their own pair
        // again costs the first cache 64 lines a step and the kernel 40 %, measured)
        tn = strip_pair(sp, sh ? (nx < maxdw ? nx : maxdw) : 0u);
        pend = sh;
    }
#define PZG_SR(f) PZG_LV(rd.f, k)
    // one lane's token at its position: tb = its bits (>= 128: a stopper), tk = the token
    // Round 6 -- two literals a step: three tokens in five of text are literals (all but one in a hundred of literal-heavy data), and
    // the token BEHIND a literal starts where the distance code of a match would: the same stream bits index the literal/length table
    // as index the distance table, in the same trip to the LDS.  e2 = that entry (meaningful when tk is a literal; taken when it is a
    // literal of the primary table itself: 6,190 steps for the 8,440 tokens of a 32 KiB text stream, 16,260 for the 31,880 of a
    // literal-heavy one)
    template <bool FX>
    PZG_FN void strip_token(uint64_t w0, uint64_t w1, uint32_t r, bool lsub, bool dsub, uint32_t &tb, uint32_t &tk, uint32_t &e2)
    {
        const uint32_t b0 = (uint32_t)w0, b1 = (uint32_t)(w0 >> 32), b2 = (uint32_t)w1, b3 = (uint32_t)(w1 >> 32);
        const bool up = r >= 32u;
        Spec t;
        spec_bits<FX>(t, up ? b1 : b0, up ? b2 : b1, up ? b3 : b2, r);  // (the funnel shifts take r modulo 32)
        if (!FX && lsub) spec_sub(t);  // (wave-uniform)
        spec_dist<FX>(t);
        e2 = lit_table<FX>()[t.w2 & ((1u << lit_bits<FX>()) - 1u)];
        if (!FX && dsub) spec_dsub(t);
        spec_finish(t, tb, tk);
    }
    PZG_FN static bool entry_is_literal(uint32_t e) { return (e & (ENT_MATCH | ENT_STOP)) == 0u; }
    // One step of phase A for every lane still in its run-up; false: none is.  (T, PD: this step's landing register.)
    template <bool FX>
    PZG_FN bool strip_step_a(const uint32_t *sp, uint32_t maxdw, bool lsub, bool dsub, StripReader &rd, LaneVec<uint32_t> &P,
                             const LaneVec<uint32_t> &LIM, LaneVec<uint32_t> &CNT)
    {
        LaneVec<bool> ACT;
        PZG_LANES_BEGIN(k)
            PZG_LV(ACT, k) = PZG_LV(P, k) < PZG_LV(LIM, k);
        PZG_LANES_END
        if (lanes_ballot(ACT) == 0ull) return false;
        PZG_MARK("sa.begin");
        PZG_LANES_BEGIN(k)
            strip_top(sp, maxdw, PZG_SR(W0), PZG_SR(W1), PZG_SR(T), PZG_SR(TN), PZG_SR(R), PZG_SR(NX), PZG_SR(PEND));
            uint32_t tb, tk, e2;
            strip_token<FX>(PZG_SR(W0), PZG_SR(W1), PZG_SR(R), lsub, dsub, tb, tk, e2);
            // (a second literal, if it starts in front of the strip: the strip's first token is the strip's)
            const bool two = PZG_LV(ACT, k) & (tb < 128u) & ((int32_t)tk >= 0) & entry_is_literal(e2) & (PZG_LV(P, k) + tb < PZG_LV(LIM, k));
            const uint32_t adv = PZG_LV(ACT, k) ? (tb < 128u ? tb + (two ? e2 & 31u : 0u) : 1u) : 0u;  // (no token here: this is not the chain yet)
            PZG_LV(CNT, k) += PZG_LV(ACT, k) ? (two ? 2u : 1u) : 0u;  // (the run-up's tokens: strip_profile_check)
            PZG_LV(P, k) += adv;
            PZG_SR(R) += adv;
        PZG_LANES_END
        PZG_MARK("sa.end");
        PZG_STAT(16, 1);  // steps of phase A
        return true;
    }
    // what a lane of phase B has produced so far
    struct SeqOut {
        LaneVec<uint32_t> STF;           // 1 = met a stopper
        LaneVec<uint32_t> OB;            // (resumable instance) the output bytes of the lane's tokens
        LaneVec<uint32_t> NR, NLB, LR;   // records / literal bytes produced; literals since the last record
        LaneVec<uint32_t> REC[SEQ_G];    // the last records, REC[SEQ_G - 1] the newest
        LaneVec<uint32_t> LA[4];         // the last 16 literal bytes, the newest in the top byte of LA[3]
    };
    PZG_FN void seq_store_records(const SeqOut &o, uint32_t k, uint32_t at)  // (dword index)
    {
        uint32_t *q = strip + at;
#pragma unroll
        for (uint32_t g = 0; g < SEQ_G; ++g) q[g] = PZG_LV(o.REC[g], k);
    }
    PZG_FN void seq_store_lits(const SeqOut &o, uint32_t k, uint32_t at)  // (byte offset, a multiple of 4)
    {
        uint32_t *q = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(strip) + at);
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) q[g] = PZG_LV(o.LA[g], k);
    }
    // ... of phase B, for the lanes of `dirty` that have not reached the end of their strip
    // INJ (strip_span, codes whose long symbols are in constant use): a step of the lanes of `dirty` alone -- lanes that stood at a token
    // the tables do not resolve --, each with the token (TBX bits, TKX) that decode_long() found there
    template <bool FX, bool INJ = false>
    PZG_FN bool strip_step_b(const uint32_t *sp, uint32_t maxdw, bool lsub, bool dsub, uint64_t dirty, StripReader &rd, SeqOut &o,
                             LaneVec<uint32_t> &P, const LaneVec<uint32_t> &LIM, const LaneVec<uint32_t> *TBX = nullptr,
                             const LaneVec<uint32_t> *TKX = nullptr)
    {
        LaneVec<bool> ACT;
        PZG_LANES_BEGIN(k)
            // (a lane whose literal area is full stands where it is: "out of steps", see strip_span -- a step takes up to two literals)
            PZG_LV(ACT, k) = lane_bit(dirty, k) & (PZG_LV(o.STF, k) == 0u) & (PZG_LV(P, k) < PZG_LV(LIM, k)) & (PZG_LV(o.NLB, k) < STRIP_TMAX);
        PZG_LANES_END
        if (lanes_ballot(ACT) == 0ull) return false;
        if (!INJ) PZG_MARK("sb.begin");
        PZG_LANES_BEGIN(k)
            strip_top(sp, maxdw, PZG_SR(W0), PZG_SR(W1), PZG_SR(T), PZG_SR(TN), PZG_SR(R), PZG_SR(NX), PZG_SR(PEND));
            uint32_t tb, tk, e2;
            if constexpr (INJ) {
                tb = PZG_LV((*TBX), k);
                tk = PZG_LV((*TKX), k);
                e2 = ENT_STOP;  // (no second literal)
            } else {
                strip_token<FX>(PZG_SR(W0), PZG_SR(W1), PZG_SR(R), lsub, dsub, tb, tk, e2);
            }
            const bool act = PZG_LV(ACT, k), stop = tb >= 128u;
            const bool ok = act & !stop;
            PZG_LV(o.STF, k) = (act & stop) ? 1u : PZG_LV(o.STF, k);
            const bool is_m = ok & ((int32_t)tk < 0), is_l = ok & ((int32_t)tk >= 0);
            // the literal behind a literal goes with it -- if it starts inside the strip (the next strip's first token is the next
            // lane's), if the two do not straddle a group of 16 literal bytes (a group is stored when its last byte comes), and while
            // the literal area has room for both
            const uint32_t nlb0 = PZG_LV(o.NLB, k);
            const bool two = is_l & entry_is_literal(e2) & (PZG_LV(P, k) + tb < PZG_LV(LIM, k)) & ((nlb0 & 15u) != 15u) & (nlb0 + 2u <= STRIP_TMAX) &
                             (!SEQ_LROVF || PZG_LV(o.LR, k) < 253u);  // (... and a run's overflow record is written at 255 literals exactly)
            const uint32_t nl = is_l ? (two ? 2u : 1u) : 0u;
            if (RES) PZG_LV(o.OB, k) += ok ? ((tk >> 16) & 511u) + (two ? 1u : 0u) : 0u;
            // a literal: its byte enters the 16-byte accumulator from the top (a funnel shift by 8, 16 or nothing: no selects)
            const uint32_t sh = nl << 3;
            PZG_LV(o.LA[0], k) = funnel(PZG_LV(o.LA[1], k), PZG_LV(o.LA[0], k), sh);
            PZG_LV(o.LA[1], k) = funnel(PZG_LV(o.LA[2], k), PZG_LV(o.LA[1], k), sh);
            PZG_LV(o.LA[2], k) = funnel(PZG_LV(o.LA[3], k), PZG_LV(o.LA[2], k), sh);
            PZG_LV(o.LA[3], k) = funnel(((tk >> 8) & 0xffu) | (e2 & 0xff00u), PZG_LV(o.LA[3], k), sh);
            const uint32_t nlb = nlb0 + nl, lr = PZG_LV(o.LR, k) + nl;
            PZG_LV(o.NLB, k) = nlb;
            // a record: a match closes the sequence its literals opened
            const bool ovf = SEQ_LROVF && (is_l & (lr == 255u));  // (the record of a literal run that is full: no match, whatever its low bits say)
            const bool emit = is_m | ovf;
            const uint32_t a = tk - 0x00030001u;  // (length - 3) << 16 | distance - 1: no borrow (distance >= 1)
            const uint32_t rec = (a & 0x7fffu) | (((a >> 16) & 0xffu) << 15) | (lr << 23) | (ovf ? SEQ_NOMATCH : 0u);
#pragma unroll
            for (uint32_t g = 0; g + 1u < SEQ_G; ++g) PZG_LV(o.REC[g], k) = emit ? PZG_LV(o.REC[g + 1u], k) : PZG_LV(o.REC[g], k);
            PZG_LV(o.REC[SEQ_G - 1u], k) = emit ? rec : PZG_LV(o.REC[SEQ_G - 1u], k);
            const uint32_t nr = PZG_LV(o.NR, k) + (emit ? 1u : 0u);
            PZG_LV(o.NR, k) = nr;
            PZG_LV(o.LR, k) = emit ? 0u : lr;
            const uint32_t adv = ok ? tb + (two ? e2 & 31u : 0u) : 0u;
            PZG_LV(P, k) += adv;
            PZG_SR(R) += adv;
            // full groups: one aligned store per lane that has one.  (Round 4, measured on text / literal-heavy data with one dword
            // per token: groups of 4 / 8 / 16 tokens 217 / 234 / 237 and 93 / 113 / 120 GiB/s -- the scratch is written in pieces of
            // lines, and the fewer and larger the pieces the less of it is written twice.)
            if (is_l & ((nlb & 15u) == 0u)) seq_store_lits(o, k, reg_lit(k) + nlb - 16u);
            if (emit & ((nr & (SEQ_G - 1u)) == 0u)) seq_store_records(o, k, reg_rec(k) + nr - SEQ_G);
        PZG_LANES_END
        if (!INJ) PZG_MARK("sb.end");
        PZG_STAT(17, 1);  // steps of phase B
        return true;
    }
    // the 64 stream bits at lane k1's position (a lane that has not moved since its last strip_top: R < 64)
    PZG_FN static uint64_t strip_peek64(const StripReader &rd, uint32_t k1)
    {
        const uint32_t r = lane_get(rd.R, k1);
        const uint64_t w0 = lane_get64(rd.W0, k1), w1 = lane_get64(rd.W1, k1);
        return r == 0u ? w0 : (w0 >> r) | (w1 << (64u - r));
    }
    // The token at the head of the 64 stream bits w, if it is one that only decode_long() resolves (a literal, or a match whose length or
    // distance code is long); false: anything else -- the lane stays stopped and the span ends there as before.  Wave-uniform, as
    // token_step_checked() is; nothing is checked against the stream's end here: a lane's position is verified like every other.
    PZG_FN bool strip_resolve(uint64_t w, uint32_t &tb, uint32_t &tk)
    {
        uint32_t bits = (uint32_t)w;
        bool lng = false;
        uint32_t e = uni(L.lit_lut[bits & ((1u << LIT_BITS) - 1u)]);
        uint32_t kind = ent_kind_lit(e);
        if (kind == K_SUB) {
            e = uni(L.sub[ent_sub_index(e) + ((bits >> LIT_BITS) & ((1u << ent_n(e)) - 1u))]);
            kind = ent_kind_lit(e);
        }
        if (kind == K_LONG) {
            e = decode_long<TREE_LITLEN>(bits, &L.lit_meta, L.lens, lit_n, lit_e15);
            kind = ent_kind_lit(e);
            lng = true;
        }
        if (kind == K_LIT) {
            tb = ent_n(e);
            tk = e;
            return lng;
        }
        if (kind != K_BASE) return false;
        const uint32_t n = ent_base_n(e), tot = ent_base_tot(e);
        const uint32_t len = ent_len_base(e) + ((bits >> n) & ((1u << (tot - n)) - 1u));
        bits = (uint32_t)(w >> tot);  // (tot <= 20: 44 bits are left, a distance takes 28 at the most)
        uint32_t d = uni(L.dist_lut[bits & ((1u << DIST_BITS) - 1u)]);
        uint32_t dk = ent_kind_dist(d);
        if (dk == K_SUB) {
            d = uni(L.sub[ent_sub_index(d) + ((bits >> DIST_BITS) & ((1u << ent_n(d)) - 1u))]);
            dk = ent_kind_dist(d);
        }
        if (dk == K_LONG) {
            d = decode_long<TREE_DIST>(bits, &L.dist_meta, L.lens + lit_n, dist_n, dist_e15);
            dk = ent_kind_dist(d);
            lng = true;
        }
        if (dk != K_BASE || !lng) return false;
        const uint32_t dn = ent_base_n(d), dtot = ent_base_tot(d);
        const uint32_t dist = ent_val(d) + ((bits >> dn) & ((1u << (dtot - dn)) - 1u));
        tb = tot + dtot;
        tk = TK_MATCH | (len << 16) | dist;
        return true;
    }
    // a lane that ran has reached the end of its strip: the literals it ended in become a last record (no match), and its last,
    // partial groups are stored -- the last SEQ_G records / 16 literals wherever they end (they may reach into the slack below)
    PZG_FN void strip_finish_lane(SeqOut &o, uint32_t k)
    {
        const uint32_t lr = PZG_LV(o.LR, k);
        const bool fin = lr != 0u;
#pragma unroll
        for (uint32_t g = 0; g + 1u < SEQ_G; ++g) PZG_LV(o.REC[g], k) = fin ? PZG_LV(o.REC[g + 1u], k) : PZG_LV(o.REC[g], k);
        PZG_LV(o.REC[SEQ_G - 1u], k) = fin ? (SEQ_NOMATCH | (lr << 23)) : PZG_LV(o.REC[SEQ_G - 1u], k);
        PZG_LV(o.NR, k) += fin ? 1u : 0u;
        PZG_LV(o.LR, k) = 0u;
        seq_store_records(o, k, reg_rec(k) + PZG_LV(o.NR, k) - SEQ_G);
        // (the literal store is made of aligned dwords: the accumulator moves down by the 0-3 bytes that round the count up)
        const uint32_t nlb = PZG_LV(o.NLB, k), sh = ((0u - nlb) & 3u) << 3;
        PZG_LV(o.LA[0], k) = funnel(PZG_LV(o.LA[1], k), PZG_LV(o.LA[0], k), sh);
        PZG_LV(o.LA[1], k) = funnel(PZG_LV(o.LA[2], k), PZG_LV(o.LA[1], k), sh);
        PZG_LV(o.LA[2], k) = funnel(PZG_LV(o.LA[3], k), PZG_LV(o.LA[2], k), sh);
        PZG_LV(o.LA[3], k) = funnel(0u, PZG_LV(o.LA[3], k), sh);
        seq_store_lits(o, k, reg_lit(k) + ((nlb + 3u) & ~3u) - 16u);
    }
    // ---- the wave's profile: strips of equal WORK ---------------------------------------------------------------------------
    // Phase B takes as many steps as the strip with the most tokens has tokens, and equal strips are far from equal work: a
    // stream's first tokens are literals (there is nothing to match yet) and its matches grow as the window fills -- measured on
    // the 32 KiB text streams of the headline batch, 64 equal strips hold 232 tokens in the first and ~105 in the last (mean 131):
    // the lanes of phase B were busy 56 % of the time.  How the tokens are spread over a stream is much the same from one stream
    // of a batch to the next, so a wave REMEMBERS it: after the first span of a stream, the bit positions (from the span's start)
    // at which 0/64, 1/64, ... 63/64 of the span's tokens had gone by -- 64 dwords, the span's extent, its token count and a
    // magic word, in the wave's scratch (it outlives streams and launches like the wave's LDS does; there is no LDS left for it).
    // The first span of the next stream cuts its strips at those positions (stretched or cut to the stream's own length; behind
    // the profile's end the last density goes on), as long a span as PROF_TGT tokens per lane allow.  Nothing about the result
    // depends on it (every lane's start is verified as before, a lane out of steps ends the span and forgets the profile); spans
    // further into a stream, where the density is flat, keep equal strips.
    // a / b, roughly (both below 2^27; b > 0): the profile needs no exact quotients
    PZG_FN static uint32_t prof_div(uint32_t a, uint32_t b)
    {
#if PZG_DEVICE_PASS
        return (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
#else
        return (uint32_t)((float)a / (float)b);
#endif
    }
    // a constant that is not worth a register of its own for the kernel's whole life: these functions run once per stream, and
    // what the compiler finds constant in them it would otherwise move to the kernel's first lines and keep (see take_prefetch)
    PZG_FN static uint32_t prof_k(uint32_t c)
    {
#if PZG_DEVICE_PASS
        asm volatile("" : "+s"(c));
#endif
        return c;
    }
    // ... and a lane number that is not worth one either: the address of "the profile's dword of this lane" is the same for the
    // kernel's whole life, and would be computed in its first lines and kept (spilled) until the one place that uses it
    PZG_FN static uint32_t prof_lane(uint32_t k)
    {
#if PZG_DEVICE_PASS
        asm volatile("" : "+v"(k));
#endif
        return k;
    }
    PZG_FN bool strip_profile_layout(uint64_t cav64, LaneVec<uint32_t> &LO, uint32_t &xspan)
    {
        // (every lane computes the span's few common values for itself: as wave-uniform values they would need some twenty scalar
        // registers where none is free, and cost the kernel a vector register for their spills)
        const uint32_t xav = 64u * (cav64 > 4096u ? 4096u : (uint32_t)cav64);
        LaneVec<uint32_t> Q, QX, QT, Q63, J, A, B, Q8, WL;
        LaneVec<bool> LE, BADP;
        PZG_LANES_BEGIN(k)
            PZG_LV(Q, k) = strip_load(PROF_OFF + prof_lane(k));
            PZG_LV(QX, k) = strip_load(PROF_OFF + 64u);
            PZG_LV(QT, k) = strip_load(PROF_OFF + 65u);
            PZG_LV(BADP, k) = strip_load(PROF_OFF + 66u) != PROF_MAGIC;
            PZG_LV(J, k) = strip_load(PROF_OFF + 67u);  // streams for which the profile is not to be consulted (strip_profile_check)
        PZG_LANES_END
        if (lanes_ballot(BADP) != 0ull) {  // no profile yet (the scratch is as the allocator left it)
            const uint32_t zero = prof_k(0u);  // (see prof_k: a pair of zeros in registers from the kernel's first line on, otherwise)
            // (the profile's few common words are stored by every lane alike -- the same value to the same address -- never under a
            // test of the lane number: ONE lane-dependent branch in this code made the compiler structurize the region around it, which
            // reaches as far as the windows' loop: 23 % more vector and 26 % more scalar instructions for a 2 KiB stream, measured)
            strip[PROF_OFF + 67u] = zero;
            strip[PROF_OFF + 68u] = zero;
            return false;
        }
        PZG_LANES_BEGIN(k)
            PZG_LV(BADP, k) = PZG_LV(J, k) != 0u;
        PZG_LANES_END
        if (lanes_ballot(BADP) != 0ull) {
            const uint32_t left = lane_get(J, 0u) - 1u;
            strip[PROF_OFF + 67u] = left;
            return false;
        }
        PZG_LANES_BEGIN(k)
            PZG_LV(J, k) = 63u;
        PZG_LANES_END
        lanes_gather(Q63, Q, J);
        // (the quantiles must not decrease: the interpolation below takes differences of neighbours, and the scratch outlives streams
        // and launches -- a well-marked profile whose words are not one is never used: VERDICT r5 item 5)
        LaneVec<uint32_t> QP;
        PZG_LANES_BEGIN(k)
            PZG_LV(J, k) = k - 1u;
        PZG_LANES_END
        lanes_gather(QP, Q, J);
        PZG_LANES_BEGIN(k)
            const uint32_t qt = PZG_LV(QT, k), qx = PZG_LV(QX, k);
            PZG_LV(BADP, k) = (qt < 64u) | (qt > prof_k(64u * STRIP_TMAX)) | (qx > prof_k(1u << 18)) | (PZG_LV(Q63, k) >= qx) | ((k == 0u) & (PZG_LV(Q, k) != 0u)) |
                              ((k != 0u) & (PZG_LV(Q, k) < PZG_LV(QP, k)));
            PZG_LV(LE, k) = PZG_LV(Q, k) <= xav;
        PZG_LANES_END
        if (lanes_ballot(BADP) != 0ull) return false;
        const uint32_t n = popc64(lanes_ballot(LE));  // quantiles at or below what the stream has left (1 or more: quantile 0 is bit 0)
        PZG_LANES_BEGIN(k)
            PZG_LV(J, k) = n - 1u;
        PZG_LANES_END
        lanes_gather(A, Q, J);
        PZG_LANES_BEGIN(k)
            PZG_LV(J, k) = n;
        PZG_LANES_END
        lanes_gather(B, Q, J);
        // the span's extent in quantiles (q8: 1/256ths): what the stream has left, or what PROF_TGT tokens per lane allow
        PZG_LANES_BEGIN(k)
            const uint32_t qx = PZG_LV(QX, k);
            const uint32_t k8191 = prof_k(8191u), kmax = prof_k(448u << 8), k64 = prof_k(64u << 8), ktgt = prof_k(PROF_TGT << 20);
            uint32_t wl = qx - PZG_LV(Q63, k);  // the last quantile's bits: what a quantile behind the profile's end is taken to be
            wl = wl > k8191 ? k8191 : wl;
            PZG_LV(WL, k) = wl;
            const uint32_t a = PZG_LV(A, k), b = n < 64u ? PZG_LV(B, k) : qx;
            const bool past = xav >= qx;
            const uint32_t num = past ? xav - qx : xav - a, den = past ? wl : (b > a ? b - a : 1u);
            uint32_t fr = prof_div(num << 8, den);
            fr = fr > kmax ? kmax : fr;
            uint32_t q8 = (past ? k64 : ((n - 1u) << 8)) + fr;
            const uint32_t qcap = prof_div(ktgt, PZG_LV(QT, k));  // QT x q8 / (64 x 256) tokens over 64 lanes
            PZG_LV(Q8, k) = q8 > qcap ? qcap : q8;
        PZG_LANES_END
        // lane k's strip begins at quantile k q / 64 (between two entries: in proportion); the span ends at quantile q
        PZG_LANES_BEGIN(k)
            const uint32_t i0 = (k * PZG_LV(Q8, k)) >> 14;
            PZG_LV(J, k) = i0 < 63u ? i0 : 63u;
        PZG_LANES_END
        lanes_gather(A, Q, J);
        PZG_LANES_BEGIN(k)
            PZG_LV(J, k) += 1u;
        PZG_LANES_END
        lanes_gather(B, Q, J);
        PZG_LANES_BEGIN(k)
            const uint32_t qx = PZG_LV(QX, k), wl = PZG_LV(WL, k);
            const uint32_t x0 = (k * PZG_LV(Q8, k)) >> 6, i0 = x0 >> 8;
            const uint32_t a0 = PZG_LV(A, k), a1 = i0 >= 63u ? qx : PZG_LV(B, k);
            const uint32_t lo = i0 < 64u ? a0 + (((x0 & 255u) * (a1 - a0)) >> 8) : qx + (((x0 - prof_k(64u << 8)) * wl) >> 8);
            PZG_LV(LO, k) = lo < xav ? lo : xav;
        PZG_LANES_END
        // (the end: what lane "64" would begin at)
        PZG_LANES_BEGIN(k)
            const uint32_t i1 = PZG_LV(Q8, k) >> 8;
            PZG_LV(J, k) = i1 < 63u ? i1 : 63u;
        PZG_LANES_END
        lanes_gather(A, Q, J);
        PZG_LANES_BEGIN(k)
            PZG_LV(J, k) += 1u;
        PZG_LANES_END
        lanes_gather(B, Q, J);
        LaneVec<uint32_t> XE;
        PZG_LANES_BEGIN(k)
            const uint32_t qx = PZG_LV(QX, k), wl = PZG_LV(WL, k), q8 = PZG_LV(Q8, k), i1 = q8 >> 8;
            const uint32_t e0 = PZG_LV(A, k), e1 = i1 >= 63u ? qx : PZG_LV(B, k);
            const uint32_t xe = i1 < 64u ? e0 + (((q8 & 255u) * (e1 - e0)) >> 8) : qx + (((q8 - prof_k(64u << 8)) * wl) >> 8);
            PZG_LV(XE, k) = xe < xav ? xe : xav;
        PZG_LANES_END
        const uint32_t xe = lane_get(XE, 0u);
        if (xe < 64u * STRIP_CMIN) return false;
        xspan = xe;
        return true;
    }
    // The profile is only as good as the last stream resembles this one.  A batch of one kind is the rule -- but laid out by the
    // profile of ANOTHER kind of stream, a span can be far worse than equal strips (measured on a batch that alternates text, html,
    // literal-heavy and binary streams of five sizes: 17.8 ms with the profile trusted blindly, 11.6 ms without one): the strips of
    // a text stream's decaying profile give the last lanes of a flat stream twice their share, or more tokens than a lane has steps.
    // The run-ups know: each lane has just decoded the `back` bits in front of its strip and counted the tokens, so its strip of
    // w bits will hold about cnt w / back of them.  If the mean of these estimates is more than a lane has steps for (this
    // stream's tokens are shorter than the profile's), or one of them twice the mean, or the largest clearly more than the
    // busiest of 64 EQUAL strips would get by the same counts, the span is laid out again in equal strips (a second phase A: ~5 %
    // of the stream's time), and the profile is not consulted for the next few streams -- twice as many after every failure in a row.
    PZG_FN bool strip_profile_check(const LaneVec<uint32_t> &CNT, const LaneVec<uint32_t> &LIM, uint32_t r0, uint32_t back, uint32_t xspan)
    {
        LaneVec<uint32_t> NEXT, LN, EST, ONE, E1, MX, SM, UX;
        PZG_LANES_BEGIN(k)
            PZG_LV(NEXT, k) = k + 1u;
            PZG_LV(ONE, k) = 1u;
        PZG_LANES_END
        lanes_gather(LN, LIM, NEXT);
        PZG_LANES_BEGIN(k)
            const uint32_t lim = PZG_LV(LIM, k), lo = lim - r0, w = (k == 63u ? r0 + xspan : PZG_LV(LN, k)) - lim;
            const uint32_t rb = lo > back ? back : lo;  // the run-up's bits (none in lane 0, few in the lanes next to it)
            PZG_LV(EST, k) = rb >= 64u ? prof_div(PZG_LV(CNT, k) * w, rb) : 0u;
            PZG_LV(UX, k) = rb >= 64u ? prof_div(PZG_LV(CNT, k) * (xspan >> 6), rb) : 0u;  // ... and an equal strip around here
        PZG_LANES_END
        lanes_gather(E1, EST, ONE);  // (lanes without an estimate of their own: the first lane that has one is not far)
        PZG_LANES_BEGIN(k)
            PZG_LV(MX, k) = PZG_LV(EST, k);
            PZG_LV(SM, k) = PZG_LV(EST, k) != 0u ? PZG_LV(EST, k) : PZG_LV(E1, k);
        PZG_LANES_END
        lanes_iscan_max(MX);
        lanes_iscan_max(UX);
        lanes_iscan_add(SM);
        const uint32_t mx = lane_get(MX, 63u), sm = lane_get(SM, 63u), ux = lane_get(UX, 63u);
#ifdef PZG_DBG_CHECK
        fprintf(stderr, "check mx=%u mean=%.1f back=%u est:", mx, sm / 64.0, back);
        for (uint32_t k = 0; k < 64u; ++k) fprintf(stderr, " %u", PZG_LV(EST, k));
        fprintf(stderr, "\n");
#endif
        // (the estimates scatter by +-20 %: the mean says whether the lanes have steps enough, the largest whether one lane has far more than its share)
        // and whether equal strips would not do better by the same estimates (the busiest of them: the span's extent / 64 at the densest run-up)
        const bool ok = sm <= 64u * (STRIP_TMAX * 13u / 16u) && mx * 64u <= sm * 2u && mx * 8u <= ux * 9u;
        {   // (every lane alike: see strip_profile_layout)
            const uint32_t level = uni(strip_load(PROF_OFF + 68u));
            const uint32_t nl = ok ? 0u : (level >= 32u ? 64u : 2u * level + 2u);
            strip[PROF_OFF + 68u] = nl;
            strip[PROF_OFF + 67u] = nl;
        }
        return ok;
    }
    // ... and what the span that was just decoded teaches: lanes 0 .. last hold NR + NLB tokens each, from LO to HI (the last
    // one to `xend`).  `ok`: every lane's start was verified and none ran out of steps -- anything else forgets the profile.
    // (LIM - r0: where the strips end; a strip begins where the one in front of it ends)
    PZG_FN void strip_profile_learn(const SeqOut &o, const LaneVec<uint32_t> &LIM, uint32_t r0, uint32_t last, uint32_t xend, bool ok)
    {
        LaneVec<uint32_t> TK, TI, CNT, HX, LO, PREV;
        PZG_LANES_BEGIN(k)
            PZG_LV(TK, k) = k <= last ? PZG_LV(o.NR, k) + PZG_LV(o.NLB, k) : 0u;
            PZG_LV(TI, k) = PZG_LV(TK, k);
            PZG_LV(HX, k) = k == last ? xend : PZG_LV(LIM, k) - r0;
            PZG_LV(CNT, k) = 0u;
            PZG_LV(PREV, k) = k - 1u;
        PZG_LANES_END
        lanes_gather(LO, HX, PREV);
        PZG_LANES_BEGIN(k)
            PZG_LV(LO, k) = k == 0u ? 0u : PZG_LV(LO, k);
        PZG_LANES_END
        lanes_iscan_add(TI);
        const uint32_t T = lane_get(TI, 63u);
        if (!ok || T < 64u || xend < 64u * STRIP_CMIN / 2u) {
            const uint32_t zero = prof_k(0u);
            // (`ok` is the same in every lane.  A token count of zero makes the profile one that strip_profile_layout() rejects until
            // the next sound span has rewritten it; the magic word and the words behind it -- the back-off, PZG_OPT_PROFILE's switch
            // -- stay as they are)
            if (!ok) strip[PROF_OFF + 65u] = zero;
            return;
        }
        // quantile j lies in the first strip i whose tokens, added up, exceed j/64 of all of them
        for (uint32_t i = 0; i < 64u; ++i) {
            const uint32_t ti64 = lane_get(TI, i) << 6;
            PZG_LANES_BEGIN(j)
                PZG_LV(CNT, j) += ti64 <= j * T ? 1u : 0u;
            PZG_LANES_END
        }
        LaneVec<uint32_t> GT, GI, GL, GH;
        lanes_gather(GT, TK, CNT);
        lanes_gather(GI, TI, CNT);
        lanes_gather(GL, LO, CNT);
        lanes_gather(GH, HX, CNT);
        PZG_LANES_BEGIN(j)
            const uint32_t t = PZG_LV(GT, j), te = PZG_LV(GI, j) - t, lo = PZG_LV(GL, j), hi = PZG_LV(GH, j);
            const uint32_t num = j * T - (te << 6);  // (below 64 t)
            // (num / 64 t of the strip's bits: t <= STRIP_TMAX, the strip below 2^18 bits -- in 1/256ths of a token)
            const uint32_t q = lo + ((prof_div(num << 2, t ? t : 1u) * (hi > lo ? hi - lo : 0u)) >> 8);
            strip[PROF_OFF + prof_lane(j)] = j == 0u ? 0u : q;
        PZG_LANES_END
        strip[PROF_OFF + 64u] = xend;
        strip[PROF_OFF + 65u] = T;
        strip[PROF_OFF + 66u] = prof_k(PROF_MAGIC);
    }

    // Decodes and emits one span.  STRIP_NA: nothing done (too little input ahead; the windows take over);
    // ST_OK: the span's tokens are all out, the cursor stands behind them -- at a stopper if `stopper`; else an error status.
    // `poor` is set when the span ended after few strips because the guesses kept failing.
    template <bool FX>
    PZG_FN int strip_span(bool &stopper, bool &poor, bool &tight)
    {
        stopper = false;
        const int64_t av = br.avail();
        if (av < (int64_t)(64u * STRIP_CMIN + 192u)) return STRIP_NA;
        // (resumable instance, ADVICE r5: a span is cut to the call's room only AFTER it has been decoded -- with room for a few strips'
        // output or less the run-up and phase B would be thrown away call after call: the windows produce what still fits)
        if (RES && cap < op + 1024u + 4096u) return STRIP_NA;
        // A lane decodes at most STRIP_TMAX tokens (its region holds no more; phase B counts its steps).  No strip of STRIP_TMAX x the
        // block's shortest literal/length code bits can hold more -- but that bound is far from what a strip does hold (text: codes
        // from 4 bits, 6.7 bits per token), and a 32 KiB stream took two spans and two run-ups where one does.  So the strips are as
        // long as the code's own expectation allows: a symbol of length l turns up with probability ~2^-l under the code the
        // compressor built for it, which makes sum(count[l] l 2^-l) bits the mean literal/length code (a match's extra bits and
        // distance only make tokens longer); 7/8 of STRIP_TMAX such tokens.  A lane that runs out of steps all the same ends the span
        // where it stands (nothing is lost but the strips behind it), and the block's later spans keep to the hard bound (`tight`).
        uint32_t minlen = 7u, mean15 = 8u << 15;  // (the fixed code: 7 / 8.03 bits)
        if (!FX) {
            minlen = 0u;
            mean15 = 0u;
            for (uint32_t l = 1u; l <= 15u; ++l) {
                const uint32_t c = uni(L.lit_meta.count[l]);
                if (minlen == 0u && c != 0u) minlen = l;
                mean15 += (c * l) << (15u - l);
            }
            if (minlen == 0u) minlen = 15u;
            if (mean15 > (15u << 15)) mean15 = 15u << 15;
        }
        const uint64_t cav64 = ((uint64_t)av - 192u) >> 6;  // the last strip ends 192 bits or more in front of the stream's end
        uint32_t cmax = STRIP_TMAX * minlen;
        if (!tight) {
            const uint32_t est = ((STRIP_TMAX * 7u / 8u) * (mean15 >> 7)) >> 8;
            if (est > cmax) cmax = est;
        }
        if (cmax > 4096u) cmax = 4096u;
        // what is left of the stream goes into spans of equal strips (a last span of short strips pays the same run-up as a full one)
        uint32_t C = cmax;
        if (cav64 <= cmax) C = (uint32_t)cav64;
        else if (cav64 < 8u * cmax) {
            uint32_t nsp = 2u;
            while (nsp * cmax < (uint32_t)cav64) ++nsp;
            C = ((uint32_t)cav64 + nsp - 1u) / nsp;
        }
        if (C < STRIP_CMIN) return STRIP_NA;
        while (qn != 0u) {  // (tokens a checked step queued: they go first)
            const int se = emit_segment();
            if (se) return se;
        }
        complete_pending();  // (... and the far bytes of the last segment: the groups read the ring)
        strip_kill_window_state();
        const uint64_t pos0 = br.pos();
        const uint32_t r0 = (uint32_t)pos0 & 31u;
        const uint32_t dw0 = (uint32_t)(pos0 >> 5);
        const uint32_t *sp = br.base + dw0;
        const uint32_t maxdw = (br.ndw - dw0 - 2u) & ~1u;
        const uint64_t bit0 = stream_bit_pos();
        const bool lsub = !FX && lit_sub_used != 0u, dsub = !FX && dist_sub_used != 0u;
        PZG_T0(tsa);
        // phase A: the run-up
        uint32_t back = uni(L.strip_back);
        if (back < STRIP_BACK_MIN || back > STRIP_BACK) back = STRIP_BACK;
        // (round 6) short strips: a run-up many times a strip's length is most of a small stream's work, and a guess that fails costs
        // only a second pass over strips of a few tokens -- the run-up is STRIP_BACK_PER_C strips at the most (not below the shortest one
        // the controller uses; what the wave has learned on long strips is left alone).  Measured with the strips' shortest length at 64
        // bits instead of 256 (GiB/s, 1 M x 2 KiB / 512 K x 4 KiB level-6 streams; before: 107.2 / 135.5 -- the windows' business until
        // then): no cap 126.8 / 186.0, 2 strips 129.6 / 192.1, 3 strips 133.5 / 183.8, 4 strips 130.7 / 173.4, 6 strips 120.1 / 185.6.
        const uint32_t back_cap = STRIP_BACK_PER_C * C > STRIP_BACK_MIN ? STRIP_BACK_PER_C * C : STRIP_BACK_MIN;
        const bool back_cut = C < STRIP_BACK_MIN && back > back_cap;  // (strips of 256 bits or more keep what the wave has learned: 8 KiB streams 238 -> 245)
        if (back_cut) back = back_cap;
        StripReader rd;
        LaneVec<uint32_t> P, S, LIM;
        // (short strips -- a 4 KiB stream's hold ~30 tokens -- differ by chance more than by position: no profile for them)
        // (... and none for blocks of the fixed code: level-1 output of a few KiB is what they are in practice)
        const bool head = STRIP_PROFILE && !FX && bit0 < PROF_HEAD && cav64 >= PROF_CMIN;
        uint32_t xspan = 64u * C;  // where the last strip ends
        for (bool by_profile = head && !tight;;) {
            {
                LaneVec<uint32_t> LO;
                if (!(PZG_PROF_LIKELY(by_profile) && strip_profile_layout(cav64, LO, xspan))) {
                    by_profile = false;
                    xspan = 64u * C;
                    PZG_LANES_BEGIN(k)
                        PZG_LV(LO, k) = k * C;
                    PZG_LANES_END
                }
                PZG_LANES_BEGIN(k)
                    PZG_LV(LIM, k) = r0 + PZG_LV(LO, k);
                PZG_LANES_END
            }
            LaneVec<uint32_t> CNT;
            PZG_LANES_BEGIN(k)
                const uint32_t lo = PZG_LV(LIM, k) - r0;
                const uint32_t p = r0 + (lo > back ? lo - back : 0u);  // (from the cursor itself: exact)
                PZG_LV(P, k) = p;
                PZG_LV(CNT, k) = 0u;
                strip_open(sp, maxdw, p, PZG_SR(W0), PZG_SR(W1), PZG_SR(T), PZG_SR(R), PZG_SR(NX), PZG_SR(PEND));
            PZG_LANES_END
            for (;;)
                if (!strip_step_a<FX>(sp, maxdw, lsub, dsub, rd, P, LIM, CNT)) break;
            // strips laid out by the profile: do the run-ups agree with it?  If not, once more with equal strips
            if (!PZG_PROF_LIKELY(by_profile)) break;
            if (strip_profile_check(CNT, LIM, r0, back, xspan)) {
                PZG_STAT(30, 1);  // spans laid out by the profile
                break;
            }
            PZG_STAT(31, 1);  // layouts the run-ups rejected
            by_profile = false;
        }
        // phase B: the strips, until every lane started where its neighbour ended
        PZG_HOT_ACC(8, tsa);
        PZG_T0(tsb);
        SeqOut o;
        {   // a strip ends where the next one begins
            LaneVec<uint32_t> NEXT, LN;
            PZG_LANES_BEGIN(k)
                PZG_LV(NEXT, k) = k + 1u;
            PZG_LANES_END
            lanes_gather(LN, LIM, NEXT);
            PZG_LANES_BEGIN(k)
                PZG_LV(LIM, k) = k == 63u ? r0 + xspan : PZG_LV(LN, k);
            PZG_LANES_END
        }
        PZG_LANES_BEGIN(k)
            PZG_LV(S, k) = PZG_LV(P, k);
            PZG_LV(o.STF, k) = 0u;
            PZG_LV(o.OB, k) = 0u;
            PZG_LV(o.NR, k) = PZG_LV(o.NLB, k) = PZG_LV(o.LR, k) = 0u;
#pragma unroll
            for (uint32_t g = 0; g < SEQ_G; ++g) PZG_LV(o.REC[g], k) = 0u;
#pragma unroll
            for (uint32_t g = 0; g < 4u; ++g) PZG_LV(o.LA[g], k) = 0u;
        PZG_LANES_END
        uint64_t dirty = ~0ull, stopm = 0ull;
        uint32_t last = 63u;
        bool repaired = false;
        for (uint32_t round = 0;;) {
            if (round != 0u) {  // the lanes that start again, from where their neighbour's chain arrived
                repaired = true;
                PZG_LANES_BEGIN(k)
                    if (lane_bit(dirty, k)) {
                        const uint32_t p = PZG_LV(S, k);
                        PZG_LV(P, k) = p;
                        PZG_LV(o.STF, k) = 0u;
                        PZG_LV(o.OB, k) = 0u;
                        PZG_LV(o.NR, k) = PZG_LV(o.NLB, k) = PZG_LV(o.LR, k) = 0u;
                        strip_open(sp, maxdw, p, PZG_SR(W0), PZG_SR(W1), PZG_SR(T), PZG_SR(R), PZG_SR(NX), PZG_SR(PEND));
                    }
                PZG_LANES_END
            }
            if (FX || (use_sub & 2u) == 0u) {
                for (uint32_t steps = 0; steps < STRIP_TMAX; ++steps)  // (a token a step at the most: the regions cannot overflow)
                    if (!strip_step_b<FX>(sp, maxdw, lsub, dsub, dirty, rd, o, P, LIM)) break;
            } else {
                // (round 6) A code whose long symbols are in constant use (its tables are laid out per prefix: binary-looking records, 250
                // literals of 8 to 10 bits) has a token every ~200 that the tables do not hold.  Such a token stopped its lane and ended
                // the span there -- 120 spans' run-ups per 30 KiB stream, or the block left to the windows (`poor`).  Here the lanes that
                // stop are looked at after every step: decode_long() finds the token, one step of their own takes it, they go on.
                uint64_t stay = 0ull;  // lanes that stopped at something else: an end of block, an error, a token past the input
                for (uint32_t steps = 0; steps < STRIP_TMAX; ++steps) {
                    if (!strip_step_b<FX>(sp, maxdw, lsub, dsub, dirty, rd, o, P, LIM)) break;
                    LaneVec<bool> ST1;
                    PZG_LANES_BEGIN(k)
                        PZG_LV(ST1, k) = PZG_LV(o.STF, k) == 1u;
                    PZG_LANES_END
                    uint64_t m = lanes_ballot(ST1) & dirty & ~stay;
                    if (__builtin_expect(m == 0ull, 1)) continue;
                    LaneVec<uint32_t> TBX, TKX;
                    uint64_t inj = 0ull;
                    do {
                        const uint32_t k1 = ctz64(m);
                        m &= m - 1ull;
                        uint32_t tb = 0u, tk = 0u;
                        if (strip_resolve(strip_peek64(rd, k1), tb, tk)) {
                            PZG_LANES_BEGIN(k)
                                PZG_LV(TBX, k) = k == k1 ? tb : PZG_LV(TBX, k);
                                PZG_LV(TKX, k) = k == k1 ? tk : PZG_LV(TKX, k);
                            PZG_LANES_END
                            inj |= 1ull << k1;
                        } else {
                            stay |= 1ull << k1;
                        }
                    } while (m != 0ull);
                    if (inj == 0ull) continue;
                    PZG_LANES_BEGIN(k)
                        PZG_LV(o.STF, k) = lane_bit(inj, k) ? 0u : PZG_LV(o.STF, k);
                    PZG_LANES_END
                    strip_step_b<FX, true>(sp, maxdw, lsub, dsub, inj, rd, o, P, LIM, &TBX, &TKX);
                    ++steps;  // (the lanes of inj have taken a step more than the loop counts)
                }
            }
            PZG_LANES_BEGIN(k)
                if (lane_bit(dirty, k)) {
                    // out of steps in front of the end of its strip: the span ends where the lane stands, as at a stopper
                    if ((PZG_LV(o.STF, k) == 0u) & (PZG_LV(P, k) < PZG_LV(LIM, k))) PZG_LV(o.STF, k) = 2u;
                    strip_finish_lane(o, k);
                }
            PZG_LANES_END
            // lane k must have started where lane k - 1's chain left its strip; lanes behind the first one that stopped do not count
            LaneVec<uint32_t> NS, PREV;
            LaneVec<bool> BAD, STOPPED;
            PZG_LANES_BEGIN(k)
                PZG_LV(PREV, k) = k - 1u;
                PZG_LV(STOPPED, k) = PZG_LV(o.STF, k) != 0u;
            PZG_LANES_END
            lanes_gather(NS, P, PREV);
            stopm = lanes_ballot(STOPPED);
            last = stopm ? ctz64(stopm) : 63u;
            PZG_LANES_BEGIN(k)
                PZG_LV(BAD, k) = (k != 0u) & (k <= last) & (PZG_LV(NS, k) != PZG_LV(S, k));
            PZG_LANES_END
            dirty = lanes_ballot(BAD);
            PZG_STAT(15, 1);  // rounds of phase B
            if (dirty == 0ull || ++round >= STRIP_ROUNDS) break;
            PZG_LANES_BEGIN(k)
                PZG_LV(S, k) = PZG_LV(BAD, k) ? PZG_LV(NS, k) : PZG_LV(S, k);
            PZG_LANES_END
        }
        {   // what the next span's run-up learns from this one: the first round's guesses all held, or not
            const uint32_t nb = !(repaired || dirty != 0ull) ? (back >= STRIP_BACK_MIN + STRIP_BACK_DOWN ? back - STRIP_BACK_DOWN : STRIP_BACK_MIN)
                                            : (back + STRIP_BACK_UP <= STRIP_BACK ? back + STRIP_BACK_UP : STRIP_BACK);
            if (!back_cut && (lane_id() == 0u || PZG_WAVE == 1u)) L.strip_back = nb;
        }
        if (dirty != 0ull) {  // still a lane that started in the wrong place: the span ends in front of it
            last = ctz64(dirty) - 1u;
            stopm = 0ull;
            // (input on which the run-ups do not find the chain -- none of the corpora has such a span -- would pay six rounds for
            // a few strips every time: the rest of the block is left to the windows)
            if (last < 16u) poor = true;
        }
        if (RES) {  // never produce past this call's room: the span ends behind the last lane whose output still fits
            LaneVec<uint32_t> OBS;
            LaneVec<bool> FITS;
            PZG_LANES_BEGIN(k)
                PZG_LV(OBS, k) = k <= last ? PZG_LV(o.OB, k) : 0u;
            PZG_LANES_END
            lanes_iscan_add(OBS);
            const uint64_t room64 = cap > op + 1024u ? cap - op - 1024u : 0u;
            const uint32_t room = room64 > 0x7fffffffull ? 0x7fffffffu : (uint32_t)room64;
            PZG_LANES_BEGIN(k)
                PZG_LV(FITS, k) = PZG_LV(OBS, k) <= room;
            PZG_LANES_END
            const uint64_t fits = lanes_ballot(FITS);  // (the sums grow with the lane: ones, then zeros)
            const uint32_t kfit = fits == ~0ull ? 64u : ctz64(~fits);
            if (kfit == 0u) return STRIP_NA;  // (nothing has moved: the windows produce what still fits and report the full room)
            if (kfit <= last) {
                last = kfit - 1u;
                stopm = 0ull;
            }
        }
        const uint32_t stf_last = stopm != 0ull ? lane_get(o.STF, last) : 0u;
        stopper = stf_last == 1u;
        // (round 6: a code whose long symbols are in constant use -- 16-byte binary records with two random bytes each: 256 literals of
        // 11-13 bits -- stops a span within its first strips at a token the tables do not resolve, every time: measured on the host
        // model, 508 spans of 94 steps for 8 records each per 30 KiB stream, 1.5 GiB/s on the device.  A span that a stopper ends within
        // its first quarter leaves the rest of the block to the windows, like one whose guesses keep failing)
        if (stopper && last < 16u) poor = true;
        if (stf_last == 2u) tight = true;
        PZG_STAT(29, stf_last == 2u ? 1 : 0);  // spans ended by a lane out of steps
        const uint32_t pend = lane_get(P, last);
        if (PZG_PROF_LIKELY(head)) strip_profile_learn(o, LIM, r0, last, pend - r0, dirty == 0ull && stf_last != 2u);
        PZG_HOT_ACC(9, tsb);
        PZG_T0(tsc);
        seq_index(o.NR, last);
        s_rd = 0u;
        s_lc = 0u;
#if defined(PZG_STATS) && !PZG_DEVICE_PASS
        {
            PZG_STAT(13, 1);                 // spans
            PZG_STAT(14, s_total);           // their records
            PZG_STAT(18, dirty != 0ull ? 1 : 0);
            PZG_STAT(19, last + 1u);         // lanes that counted
        }
#endif
        // the cursor goes behind the span
        const uint64_t endbit = bit0 + (uint64_t)(pend - r0);
        in_byte0 = endbit >> 3;
        br.start(in, in_len, in_byte0);
        br.drop((uint32_t)endbit & 7u);
        strip_fence();
        seq_refill();
        PZG_HOT_ACC(10, tsc);
        PZG_T0(tse);
        // groups of sequences for as long as records are left (the fast body in a loop of its own: see hot_loop())
        for (;;) {
            const uint32_t why = seq_hot();
            if (why == SQ_DONE) break;
            if (why == SQ_FLUSH) {
                flush_to(op & ~(uint64_t)15u);
                continue;
            }
            const int se = seq_solo();
            if (se) return se;
        }
        PZG_HOT_ACC(11, tse);
        strip_kill_window_state();
        return ST_OK;
    }
    // The windows' per-lane state holds nothing while a span runs (the queue is empty, no far byte is pending): saying so --
    // zeros in, zeros out -- frees its vector registers for the span instead of keeping four of them alive across it.
    PZG_FN void strip_kill_window_state()
    {
        PZG_LANES_BEGIN(j)
            PZG_LV(QT, j) = 0u;
            PZG_LV(pendF0, j) = 0;
            PZG_LV(pendF1, j) = 0;
        PZG_LANES_END
    }
#undef PZG_SR

    // ---- the groups: every lane copies one whole sequence -----------------------------------------------------------------
    // A group is the next (up to 64) records of the span whose output fits SEQ_GLIM bytes.  One prefix sum over (literals +
    // match length) places every sequence; the lanes then store their literal runs (from the scratch's literal areas) and copy
    // their matches inside the ring, four bytes per instruction wherever four are left: an unaligned dword read and an
    // unaligned dword write, the last dword of a run overlapping the one before it instead of a tail of single bytes.
    // What makes that legal is the order of three things, byte-serial semantics kept (OutputWindow.hs:82-101, Deflate.hs:106-120):
    //   1. all literal runs of the group (they depend on nothing);
    //   2. the matches whose source ends in front of the group's first match (H): everything they read is older than the
    //      group or one of its literals.  Matches older than the ring ("far") read the stream's flushed output instead;
    //   3. what is left -- matches that read the output of other matches of the same group -- in rounds: the first one still
    //      waiting can always go, and with it every one whose source ends in front of it.  (Text: one match in 30-50.)
    // A sequence the lanes cannot take -- a match longer than SEQ_CAP, a distance reaching in front of the output, on the
    // 32 KiB ring a source the group itself would overwrite -- ends the group in front of it and goes through seq_solo().
    // Runs that wrap around the ring's end, and matches with a distance below 4 (their dwords would read what they are
    // writing), are copied byte by byte by loops that run only when a ballot says one is there.
#ifndef PZG_SEQ_GLIM
#define PZG_SEQ_GLIM 768
#endif
#ifndef PZG_SEQ_CAP
#define PZG_SEQ_CAP 32
#endif
    static constexpr uint32_t SEQ_GLIM = PZG_SEQ_GLIM;  // output bytes of a group, at most (the ring keeps RING - SEQ_GLIM bytes of history "near")
    static constexpr uint32_t SEQ_CAP = PZG_SEQ_CAP;    // longest match a lane copies by itself
    static_assert(SEQ_GLIM + SEQ_CAP + 3u + 128u <= 1024u && SEQ_GLIM >= 255u + SEQ_CAP, "far sources end a cache line or more below `flushed`; one sequence always fits");
    enum : uint32_t { SQ_OK = 0, SQ_DONE = 1, SQ_SOLO = 2, SQ_FLUSH = 3 };

    // Ring bytes at any address, by byte operations whose offsets ride in the instructions.  (Round 5, measured: unaligned
    // ds_read_b32 / ds_write_b32 work on gfx950 -- and stall the LDS for ~50 cycles per wave-instruction,
    // SQ_LDS_UNALIGNED_STALL 7.1e9 per launch; and the compiler merges four byte accesses written in C++ back into exactly that
    // dword.)  On the device they are inline assembly that runs under a lane mask put into EXEC -- no select of a dump address,
    // no lane-dependent branch: the copies of a group cost the vector unit their compares and little else.  LDS operations
    // complete in order, so the compiler's own wait counts stay sufficient with these in between; the reads wait for their data
    // themselves.  In the one-lane host model the same functions are plain loops over the lanes of the mask.
    static constexpr uint32_t RING_OFF = (uint32_t)offsetof(WaveLds<RING_BITS>, ring);  // (the wave's LDS image starts at LDS address 0: see BitReader::dma_prefetch)
    static constexpr uint32_t DUMP_REL = (uint32_t)(offsetof(WaveLds<RING_BITS>, dump) - offsetof(WaveLds<RING_BITS>, ring));
    struct Quad {
        uint8_t b[4];
    };
    // bytes [OFF, OFF + 4) of the lanes of m0 and [OFF + 4, OFF + 8) of the lanes of m1 go from ring offset SM to ring offset DM:
    // all eight reads in flight, one wait, the writes (last byte first: see the literal runs of seq_group)
    template <uint32_t OFF>
    PZG_FN void lds_copy4x2(const LaneVec<uint32_t> &SM, const LaneVec<uint32_t> &DM, uint64_t m0, uint64_t m1)
    {
#if PZG_DEVICE_PASS
        // (round 6: the eight source bytes are read as three ALIGNED dwords and funnel-shifted into place -- three LDS instructions
        // instead of eight byte reads: the LDS pipe is what the groups run out of, and this alone was worth 317 -> 323 GiB/s on text, 252 ->
        // 262 on html; the writes stay bytes: neighbouring lanes' matches share dwords)
        static_assert(RING_OFF % 4u == 0u && OFF % 8u == 0u, "aligned dwords");
        uint32_t a, sh, d0, d1, d2, q0, q1, y0, y1;
        uint64_t sv;
        asm volatile("s_mov_b64 %9, exec\n\t"
                     "s_mov_b64 exec, %12\n\t"
                     "v_and_b32 %0, -4, %10\n\t"
                     "v_lshlrev_b32 %1, 3, %10\n\t"
                     "ds_read_b32 %2, %0 offset:%c14\n\tds_read_b32 %3, %0 offset:%c15\n\tds_read_b32 %4, %0 offset:%c16\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_alignbit_b32 %5, %3, %2, %1\n\t"
                     "v_alignbit_b32 %6, %4, %3, %1\n\t"
                     "v_lshrrev_b32 %7, 8, %5\n\t"
                     "v_lshrrev_b32 %8, 8, %6\n\t"
                     "s_mov_b64 exec, %13\n\t"
                     "ds_write_b8_d16_hi %11, %8 offset:%c24\n\tds_write_b8_d16_hi %11, %6 offset:%c23\n\tds_write_b8 %11, %8 offset:%c22\n\tds_write_b8 %11, %6 offset:%c21\n\t"
                     "s_mov_b64 exec, %12\n\t"
                     "ds_write_b8_d16_hi %11, %7 offset:%c20\n\tds_write_b8_d16_hi %11, %5 offset:%c19\n\tds_write_b8 %11, %7 offset:%c18\n\tds_write_b8 %11, %5 offset:%c17\n\t"
                     "s_mov_b64 exec, %9"
                     : "=&v"(a), "=&v"(sh), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(q0), "=&v"(q1), "=&v"(y0), "=&v"(y1), "=&s"(sv)
                     : "v"(SM.v), "v"(DM.v), "s"(m0), "s"(m1), "n"(RING_OFF + OFF), "n"(RING_OFF + OFF + 4u), "n"(RING_OFF + OFF + 8u),
                       "n"(RING_OFF + OFF), "n"(RING_OFF + OFF + 1u), "n"(RING_OFF + OFF + 2u), "n"(RING_OFF + OFF + 3u), "n"(RING_OFF + OFF + 4u),
                       "n"(RING_OFF + OFF + 5u), "n"(RING_OFF + OFF + 6u), "n"(RING_OFF + OFF + 7u)
                     : "memory");
#else
        LaneVec<Quad> A, B;
        for (uint32_t j = 0; j < 64u; ++j)  // (every read of a step precedes its writes, as on the device)
            for (uint32_t t = 0; t < 4u; ++t) {
                if ((m0 >> j) & 1u) A.v[j].b[t] = L.ring[SM.v[j] + OFF + t];
                if ((m1 >> j) & 1u) B.v[j].b[t] = L.ring[SM.v[j] + OFF + 4u + t];
            }
        for (uint32_t j = 0; j < 64u; ++j)
            for (uint32_t t = 4u; t-- > 0u;) {
                if ((m1 >> j) & 1u) L.ring[DM.v[j] + OFF + 4u + t] = B.v[j].b[t];
                if ((m0 >> j) & 1u) L.ring[DM.v[j] + OFF + t] = A.v[j].b[t];
            }
#endif
    }
    // four bytes of the lanes of m4 from ST to DT, three bytes of the lanes of m3 from S3 to D3
    PZG_FN void lds_copy_tail(const LaneVec<uint32_t> &ST, const LaneVec<uint32_t> &DT, uint64_t m4, const LaneVec<uint32_t> &S3,
                              const LaneVec<uint32_t> &D3, uint64_t m3)
    {
#if PZG_DEVICE_PASS
        // (round 6: a lane is in ONE of the two masks -- one source and one destination per lane, two aligned dwords read and
        // funnel-shifted, four byte writes: six LDS instructions where seven byte reads and seven byte writes made fourteen; with the
        // trips' dword reads, text 318 -> 330 GiB/s, html 252 -> 273)
        uint32_t sa, da, a, sh, d0, d1, q, y;
        uint64_t sv;
        asm volatile("s_mov_b64 %8, exec\n\t"
                     "s_or_b64 exec, %13, %14\n\t"
                     "v_cndmask_b32_e64 %0, %11, %9, %13\n\t"
                     "v_cndmask_b32_e64 %1, %12, %10, %13\n\t"
                     "v_and_b32 %2, -4, %0\n\t"
                     "v_lshlrev_b32 %3, 3, %0\n\t"
                     "ds_read_b32 %4, %2 offset:%c15\n\tds_read_b32 %5, %2 offset:%c19\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_alignbit_b32 %6, %5, %4, %3\n\t"
                     "v_lshrrev_b32 %7, 8, %6\n\t"
                     "s_mov_b64 exec, %13\n\t"
                     "ds_write_b8_d16_hi %1, %7 offset:%c18\n\t"
                     "s_or_b64 exec, %13, %14\n\t"
                     "ds_write_b8_d16_hi %1, %6 offset:%c17\n\tds_write_b8 %1, %7 offset:%c16\n\tds_write_b8 %1, %6 offset:%c15\n\t"
                     "s_mov_b64 exec, %8"
                     : "=&v"(sa), "=&v"(da), "=&v"(a), "=&v"(sh), "=&v"(d0), "=&v"(d1), "=&v"(q), "=&v"(y), "=&s"(sv)
                     : "v"(ST.v), "v"(DT.v), "v"(S3.v), "v"(D3.v), "s"(m4), "s"(m3), "n"(RING_OFF), "n"(RING_OFF + 1u), "n"(RING_OFF + 2u), "n"(RING_OFF + 3u),
                       "n"(RING_OFF + 4u)
                     : "memory", "scc");
#else
        LaneVec<Quad> A, B;
        for (uint32_t j = 0; j < 64u; ++j)
            for (uint32_t t = 0; t < 4u; ++t) {
                if ((m4 >> j) & 1u) A.v[j].b[t] = L.ring[ST.v[j] + t];
                if (((m3 >> j) & 1u) && t < 3u) B.v[j].b[t] = L.ring[S3.v[j] + t];
            }
        for (uint32_t j = 0; j < 64u; ++j)
            for (uint32_t t = 4u; t-- > 0u;) {
                if (((m3 >> j) & 1u) && t < 3u) L.ring[D3.v[j] + t] = B.v[j].b[t];
                if ((m4 >> j) & 1u) L.ring[DT.v[j] + t] = A.v[j].b[t];
            }
#endif
    }
    // the dword X goes to ring offset A: byte t of the lanes of m[t] (a lane's masks are nested: m[3] in m[2] in m[1] in m[0])
    PZG_FN void lds_put_bytes(const LaneVec<uint32_t> &A, const LaneVec<uint32_t> &X, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3)
    {
#if PZG_DEVICE_PASS
        uint32_t y;
        uint64_t sv;
        asm volatile("s_mov_b64 %1, exec\n\t"
                     "s_mov_b64 exec, %7\n\t"
                     "v_lshrrev_b32 %0, 8, %3\n\t"
                     "ds_write_b8_d16_hi %2, %0 offset:%c11\n\t"
                     "s_mov_b64 exec, %6\n\t"
                     "ds_write_b8_d16_hi %2, %3 offset:%c10\n\t"
                     "s_mov_b64 exec, %5\n\t"
                     "v_lshrrev_b32 %0, 8, %3\n\t"
                     "ds_write_b8 %2, %0 offset:%c9\n\t"
                     "s_mov_b64 exec, %4\n\t"
                     "ds_write_b8 %2, %3 offset:%c8\n\t"
                     "s_mov_b64 exec, %1"
                     : "=&v"(y), "=&s"(sv)
                     : "v"(A.v), "v"(X.v), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "n"(RING_OFF), "n"(RING_OFF + 1u), "n"(RING_OFF + 2u), "n"(RING_OFF + 3u)
                     : "memory");
#else
        for (uint32_t j = 0; j < 64u; ++j) {
            if ((m3 >> j) & 1u) L.ring[A.v[j] + 3u] = (uint8_t)(X.v[j] >> 24);
            if ((m2 >> j) & 1u) L.ring[A.v[j] + 2u] = (uint8_t)(X.v[j] >> 16);
            if ((m1 >> j) & 1u) L.ring[A.v[j] + 1u] = (uint8_t)(X.v[j] >> 8);
            if ((m0 >> j) & 1u) L.ring[A.v[j]] = (uint8_t)X.v[j];
        }
#endif
    }
    // ... all four bytes of the lanes of m
    PZG_FN void lds_put_dword(const LaneVec<uint32_t> &A, const LaneVec<uint32_t> &X, uint64_t m)
    {
#if PZG_DEVICE_PASS
        uint32_t y;
        uint64_t sv;
        asm volatile("s_mov_b64 %1, exec\n\t"
                     "s_mov_b64 exec, %4\n\t"
                     "v_lshrrev_b32 %0, 8, %3\n\t"
                     "ds_write_b8_d16_hi %2, %0 offset:%c8\n\tds_write_b8_d16_hi %2, %3 offset:%c7\n\tds_write_b8 %2, %0 offset:%c6\n\tds_write_b8 %2, %3 offset:%c5\n\t"
                     "s_mov_b64 exec, %1"
                     : "=&v"(y), "=&s"(sv)
                     : "v"(A.v), "v"(X.v), "s"(m), "n"(RING_OFF), "n"(RING_OFF + 1u), "n"(RING_OFF + 2u), "n"(RING_OFF + 3u)
                     : "memory");
#else
        lds_put_bytes(A, X, m, m, m, m);
#endif
    }
    // Where produced byte op + rel (rel < 0: older than the ring, flushed) lies, as an offset from far_base: in the stream's own
    // output -- or, for a resumable decoder on a small ring, in its 32 KiB history, by position modulo its size.
    PZG_FN uint32_t far_off(uint32_t op32, uint32_t fdelta, uint32_t rel) const
    {
        return RES_HIST ? (op32 + rel) & HIST_MASK : 32768u + fdelta + rel;
    }
    PZG_FN uint32_t far_load32(uint32_t off) const
    {
#if PZG_DEVICE_PASS
        if (RES_HIST)  // (the history is rewritten as the stream goes: past the L1)
            return __hip_atomic_load(reinterpret_cast<const uint32_t *>(far_base + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        uint32_t v;
        __builtin_memcpy(&v, far_base + off, 4);
        return v;
    }
    PZG_FN uint8_t far_load8(uint32_t off) const
    {
#if PZG_DEVICE_PASS
        if (RES_HIST) return __hip_atomic_load(far_base + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        return far_base[off];
    }
    static constexpr uint32_t FAR_DUMMY = RES_HIST ? 0u : FAR_IDLE;  // what a lane with no far source reads
    // The group's near matches: where they read (SM) and write (DM) in the ring, and the lanes that move dwords (norm4: four
    // bytes or more, NB whole quads from the front and one that ENDS with the match, ST -> DT -- it overlaps the one before it
    // instead of a tail of single bytes), three bytes (norm3), or go byte by byte (slow: a run that wraps around the ring's end,
    // a distance below 16 that the match is longer than).
    struct SeqCopy {
        LaneVec<uint32_t> SM, DM, LEN, NB, ST, DT;
        uint64_t norm4, norm3, slow;
    };
    template <uint32_t T>
    PZG_FN void seq_copy_trips(const SeqCopy &c, uint64_t m4)
    {
        if constexpr (T < SEQ_CAP / 8u) {
            LaneVec<bool> A0, A1;
            PZG_LANES_BEGIN(j)
                PZG_LV(A0, j) = 2u * T < PZG_LV(c.NB, j);
                PZG_LV(A1, j) = 2u * T + 1u < PZG_LV(c.NB, j);
            PZG_LANES_END
            const uint64_t m0 = m4 & lanes_ballot(A0), m1 = m4 & lanes_ballot(A1);
            if (m0 == 0ull) return;
            lds_copy4x2<8u * T>(c.SM, c.DM, m0, m1);
            PZG_STAT(26, 1);  // dword steps (of two)
            seq_copy_trips<T + 1u>(c, m4);
        }
    }
    // ... those of `rdy` are copied
    PZG_FN void seq_copy(const SeqCopy &c, uint64_t rdy)
    {
        const uint64_t m4 = rdy & c.norm4, m3 = rdy & c.norm3, sl = rdy & c.slow;
        if (m4 != 0ull) seq_copy_trips<0u>(c, m4);
        lds_copy_tail(c.ST, c.DT, m4, c.SM, c.DM, m3);
        if (__builtin_expect(sl != 0ull, 0)) {
            LaneVec<bool> ACT;
            LaneVec<uint32_t> X;
            for (uint32_t i = 0;; ++i) {
                PZG_LANES_BEGIN(j)
                    PZG_LV(ACT, j) = lane_bit(sl, j) & (i < PZG_LV(c.LEN, j));
                PZG_LANES_END
                if (lanes_ballot(ACT) == 0ull) break;
                PZG_LANES_BEGIN(j)  // (every read of a step precedes its writes, on the device and in the one-lane model alike)
                    PZG_LV(X, j) = L.ring[(PZG_LV(c.SM, j) + i) & RMASK];
                PZG_LANES_END
                PZG_LANES_BEGIN(j)
                    ring_store(PZG_LV(ACT, j), (PZG_LV(c.DM, j) + i) & RMASK, (uint8_t)PZG_LV(X, j), j);
                PZG_LANES_END
                PZG_STAT(27, 1);  // byte steps
            }
        }
    }

    // One group.  SQ_OK: emitted, the next group's records are on their way; SQ_DONE: the span is used up; SQ_SOLO: the record
    // at the head is one for seq_solo(); SQ_FLUSH: a flush is due that the fast one cannot do.  Nothing has changed in the
    // last three cases.
    // The order of the memory operations is what the group's time depends on (measured: with every load waited for where it
    // was issued a group took three trips to the L2 in a row and the kernel ran at 160 GiB/s where round 4's ran at 264):
    // as soon as the group's extent is known the NEXT group's records are asked for, then this group's literals and the far
    // matches' sources; the matches that read nothing of this group are copied while those are on their way.
    PZG_FN uint32_t seq_group()
    {
        const uint32_t n = s_qn;
        if (n == 0u) return SQ_DONE;
        if (__builtin_expect((uint32_t)(op - flushed) >= FLUSH_AT, 0)) {  // whole KiB only (the general flush goes up to op & ~15)
            const uint64_t to = flushed + ((uint32_t)(op - flushed) & ~1023u);
            if (!out_aligned() || to > cap) return SQ_FLUSH;
            flush_span<true>(to);
        }
        PZG_SEQ_T0(tq);
        PZG_MARK("g.begin");
        const uint32_t op32 = (uint32_t)op;
        uint32_t op_hi = (uint32_t)(op >> 32);
#if PZG_DEVICE_PASS
        asm("" : "+s"(op_hi));  // (opaque: see emit_body)
#endif
        const uint32_t hist = (op_hi | (op32 >> 20)) ? 0x100000u : op32 + (RING_BITS == 15 ? hist_extra : 0u);
        // ---- place the sequences: one prefix sum over (literals + match length), the literal count riding in the high half
        LaneVec<uint32_t> NL, LEN, DIST, ENDX, MO;
        PZG_LANES_BEGIN(j)
            const uint32_t rec = PZG_LV(QTN, j);
            const bool valid = j < n;
            const uint32_t nl = valid ? seq_nl(rec) : 0u, len = valid ? seq_len(rec) : 0u;
            PZG_LV(NL, j) = nl;
            PZG_LV(LEN, j) = len;
            PZG_LV(DIST, j) = seq_dist(rec);
            PZG_LV(ENDX, j) = (nl + len) | (nl << 16);
        PZG_LANES_END
        lanes_iscan_add(ENDX);  // (64 x 513 and 64 x 255: both halves stay below 2^16)
        LaneVec<bool> OVER, BAD, MIX, HASM;
        PZG_LANES_BEGIN(j)
            const uint32_t end = PZG_LV(ENDX, j) & 0xffffu, len = PZG_LV(LEN, j), m = end - len;
            PZG_LV(MO, j) = m;  // the match's first byte, relative to op (its literals end there)
            PZG_LV(OVER, j) = end > SEQ_GLIM;
            PZG_LV(HASM, j) = len != 0u;
            PZG_LV(BAD, j) = PZG_LV(DIST, j) > hist + m;  // (of a match:) reaches in front of the output: seq_solo() reports it
            // the 32 KiB ring keeps nothing else: a source the group's own bytes would overwrite first
            PZG_LV(MIX, j) = (int32_t)(m - PZG_LV(DIST, j)) < (int32_t)(SEQ_GLIM - RING);
        PZG_LANES_END
        const uint64_t stopm = lanes_ballot(OVER) | (lanes_ballot(HASM) & (lanes_ballot(BAD) | (RING_BITS == 15 ? lanes_ballot(MIX) : 0ull)));
        const uint32_t v0 = stopm ? ctz64(stopm) : n;
        const uint32_t v = v0 < n ? v0 : n;  // sequences of this group
        if (__builtin_expect(v == 0u, 0)) return SQ_SOLO;
        const uint32_t endv = lane_get(ENDX, v - 1u), run = endv & 0xffffu, lend_v = endv >> 16;
        const uint64_t taken = v >= 64u ? ~0ull : bit_field_mask(v, 0u);
        PZG_STAT(20, 1);    // groups
        PZG_STAT(21, v);    // their sequences
        PZG_STAT(22, run);  // their bytes
        // ---- the matches: where they read and write.  (Every predicate is ONE compare, balloted at once, and the classes are put
        // together on the masks by scalar instructions: a ballot of a compound predicate costs two vector instructions more.)
        SeqCopy c;
        LaneVec<uint32_t> SEND;
        LaneVec<bool> B_LEN, B_NEAR, B_WD, B_WS, B_D16, B_OV, B_D4, B_BIG, B_L4, B_L3;
        PZG_LANES_BEGIN(j)
            const uint32_t len = PZG_LV(LEN, j), dist = PZG_LV(DIST, j), m = PZG_LV(MO, j);
            const int32_t srcr = (int32_t)(m - dist);  // relative to op
            const uint32_t dm = (op32 + m) & RMASK, sm = (dm - dist) & RMASK;
            PZG_LV(c.LEN, j) = len;
            PZG_LV(c.DM, j) = dm;
            PZG_LV(c.SM, j) = sm;
            PZG_LV(c.NB, j) = (len - 1u) >> 2;
            PZG_LV(c.ST, j) = sm + len - 4u;
            PZG_LV(c.DT, j) = dm + len - 4u;
            PZG_LV(B_LEN, j) = len != 0u;
            PZG_LV(B_NEAR, j) = srcr >= (int32_t)(run - RING);  // still in the ring when the group's last byte is
            PZG_LV(B_WD, j) = dm + len > RING;
            PZG_LV(B_WS, j) = sm + len > RING;
            PZG_LV(B_D16, j) = dist < 16u;
            PZG_LV(B_OV, j) = len > dist;
            PZG_LV(B_D4, j) = dist < 4u;
            PZG_LV(B_BIG, j) = len > SEQ_CAP;  // all lanes together, when its turn comes (seq_coop)
            PZG_LV(B_L4, j) = len >= 4u;
            PZG_LV(B_L3, j) = len == 3u;
            PZG_LV(SEND, j) = (uint32_t)srcr + (len < dist ? len : dist);  // where its source ends (signed, relative to op)
        PZG_LANES_END
        const uint64_t hasm = taken & lanes_ballot(B_LEN), coopm = hasm & lanes_ballot(B_BIG);
        const uint64_t nearx = HYBRID ? hasm & lanes_ballot(B_NEAR) & ~coopm : hasm & ~coopm;
        const uint64_t wrapd = lanes_ballot(B_WD);
        const uint64_t slowx = wrapd | lanes_ballot(B_WS) | (lanes_ballot(B_D16) & (lanes_ballot(B_OV) | lanes_ballot(B_D4)));
        c.slow = nearx & slowx;
        c.norm4 = nearx & ~slowx & lanes_ballot(B_L4);
        c.norm3 = nearx & ~slowx & lanes_ballot(B_L3);
        const uint64_t nearm = nearx | coopm;
#if defined(PZG_LAB) && defined(PZG_LAB_NOFAR)  // (lab builds only: traffic attribution -- the far matches are not copied, the output is wrong)
        const uint64_t farm = 0ull;
#else
        const uint64_t farm = HYBRID ? hasm & ~nearx & ~coopm & far_okmask : 0ull;
#endif
        // ---- the far matches' sources: asked for.  They end SEQ_GLIM + ... bytes below `flushed` at the least (static_assert
        // above), in lines that are complete and final (see set_far_base).
        LaneVec<uint32_t> FO, FND, FX[4];
        uint64_t farn = farm & ~wrapd, fars = farm & wrapd;
        if (HYBRID && farm != 0ull) {
            PZG_STAT(25, 1);
            far_fence();
            const uint32_t fdelta = (uint32_t)(op - flushed);
            if (RES_HIST) {  // (a source that wraps around the history's end: byte by byte)
                LaneVec<bool> HW;
                PZG_LANES_BEGIN(j)
                    PZG_LV(HW, j) = far_off(op32, fdelta, PZG_LV(MO, j) - PZG_LV(DIST, j)) + PZG_LV(c.LEN, j) + 3u > HIST_BYTES;
                PZG_LANES_END
                const uint64_t hw = lanes_ballot(HW);
                fars = farm & (wrapd | hw);
                farn = farm & ~(wrapd | hw);
            }
            PZG_LANES_BEGIN(j)
                const uint32_t len = PZG_LV(c.LEN, j), l4 = len >= 4u ? len - 4u : 0u;
                const uint32_t nd = mask_keep(farn, j, (len + 3u) >> 2), fo = far_off(op32, fdelta, PZG_LV(MO, j) - PZG_LV(DIST, j));  // the source's first byte, from far_base
                PZG_LV(FND, j) = nd;
                PZG_LV(FO, j) = fo;
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t off = 4u * u < l4 ? 4u * u : l4;
#if PZG_DEVICE_PASS
                    PZG_LV(FX[u], j) = far_load32(u < nd ? fo + off : FAR_DUMMY);
#else
                    PZG_LV(FX[u], j) = u < nd ? far_load32(fo + off) : 0u;
#endif
                }
            PZG_LANES_END
        }
        // ---- the next group's records: asked for now, looked at when this group is done
        LaneVec<uint32_t> INFO, LOE, FLV, FLI, CL;
        PZG_LANES_BEGIN(j)
            PZG_LV(INFO, j) = PZG_LV(QINFO, j);
            PZG_LV(FLI, j) = (PZG_LV(QINFO, j) >> 8) & 63u;
            PZG_LV(LOE, j) = (PZG_LV(ENDX, j) >> 16) - PZG_LV(NL, j);  // the group's literals in front of the lane's
        PZG_LANES_END
        const uint32_t lc0 = s_lc, lit_a = reg_lit(lane_get(QINFO, 0u) & 255u);  // (lit_a: a valid place of the scratch, for the lanes that load nothing)
        s_rd += v;
        const bool at_start = seq_refill();
        lanes_gather(FLV, LOE, FLI);  // ... in front of the first lane of its region
        PZG_LANES_BEGIN(j)
            PZG_LV(CL, j) = PZG_LV(LOE, j) + PZG_LV(NL, j) - PZG_LV(FLV, j) + (((PZG_LV(INFO, j) >> 16) & 1u) ? lc0 : 0u);  // its region's literals up to and including its own
        PZG_LANES_END
        s_lc = at_start ? 0u : lane_get(CL, v - 1u);
        (void)lend_v;
        PZG_SEQ_ACC(12, tq);
        PZG_MARK("g.lits");
        // ---- this group's literals: asked for
        LaneVec<uint32_t> LAD, DL, X0, X1;
        LaneVec<bool> B_NL, B_WL, B_N8, B_N5;
        PZG_LANES_BEGIN(j)
            const uint32_t nl = PZG_LV(NL, j);
            const uint32_t la = reg_lit(PZG_LV(INFO, j) & 255u) + PZG_LV(CL, j) - nl, dl = (op32 + PZG_LV(MO, j) - nl) & RMASK;
            PZG_LV(LAD, j) = la;
            PZG_LV(DL, j) = dl;
            PZG_LV(B_NL, j) = nl != 0u;
            PZG_LV(B_WL, j) = dl + nl > RING;
            PZG_LV(B_N8, j) = nl > 8u;
            PZG_LV(B_N5, j) = nl > 4u;
        PZG_LANES_END
        const uint64_t litm = taken & lanes_ballot(B_NL), wrapl = lanes_ballot(B_WL), fastl = litm & ~wrapl, slowl = litm & wrapl;
        const uint64_t morel = fastl & lanes_ballot(B_N8), fast5 = fastl & lanes_ballot(B_N5);
        PZG_LANES_BEGIN(j)
            const uint32_t nl = PZG_LV(NL, j), la = PZG_LV(LAD, j);
            const uint32_t off1 = nl - 4u < 4u ? nl - 4u : 4u;
            PZG_LV(X0, j) = lit_load32(mask_select(fastl, j, la, lit_a));
            PZG_LV(X1, j) = lit_load32(mask_select(fast5, j, la + off1, lit_a));
        PZG_LANES_END
        PZG_SEQ_ACC(13, tq);
        PZG_MARK("g.matches");
#if defined(PZG_STATS) && !PZG_DEVICE_PASS
        {   // (lab statistics: the true depth of the group's dependencies -- rounds an exact rule would need)
            uint32_t lvl[64], maxl = 0;
            for (uint32_t j = 0; j < 64u; ++j) {
                lvl[j] = 0;
                if (!((hasm >> j) & 1ull)) continue;
                const int32_t sj = (int32_t)(PZG_LV(MO, j) - PZG_LV(DIST, j)), ej = (int32_t)PZG_LV(SEND, j);
                uint32_t l = 1;
                for (uint32_t i = 0; i < j; ++i)
                    if (((hasm >> i) & 1ull) && (int32_t)PZG_LV(MO, i) < ej && (int32_t)(PZG_LV(MO, i) + PZG_LV(LEN, i)) > sj && lvl[i] + 1u > l) l = lvl[i] + 1u;
                lvl[j] = l;
                if (l > maxl) maxl = l;
            }
            PZG_STAT(28, maxl);
        }
#endif
        // ---- 2a. the matches that read nothing of this group (most of them), while the loads are on their way
        uint64_t pend = nearm;
        if (nearm != 0ull) {
            LaneVec<bool> RDY;
            PZG_LANES_BEGIN(j)
                PZG_LV(RDY, j) = (int32_t)PZG_LV(SEND, j) <= 0;
            PZG_LANES_END
            const uint64_t rdy = nearm & lanes_ballot(RDY) & ~coopm;
            seq_copy(c, rdy);
            pend &= ~rdy;
            PZG_STAT(23, 1);  // copy rounds
        }
        PZG_SEQ_ACC(14, tq);
        // ---- 1. the literal runs: byte t of the first dword where the run has more than t bytes, the second dword ends with the run
        {
            LaneVec<bool> G2, G3, G4;
            LaneVec<uint32_t> DL1;
            PZG_LANES_BEGIN(j)
                const uint32_t nl = PZG_LV(NL, j);
                PZG_LV(G2, j) = nl >= 2u;
                PZG_LV(G3, j) = nl >= 3u;
                PZG_LV(G4, j) = nl >= 4u;
                PZG_LV(DL1, j) = PZG_LV(DL, j) + (nl - 4u < 4u ? nl - 4u : 4u);
            PZG_LANES_END
            lds_put_bytes(DL, X0, fastl, fastl & lanes_ballot(G2), fastl & lanes_ballot(G3), fastl & lanes_ballot(G4));
            lds_put_dword(DL1, X1, fast5);
        }
        if (__builtin_expect(morel != 0ull, 0)) {  // runs of more than 8 literals: two dwords a step
            LaneVec<bool> A0, A1;
            for (uint32_t i = 2u;; i += 2u) {
                PZG_LANES_BEGIN(j)
                    const uint32_t nd = (PZG_LV(NL, j) + 3u) >> 2;
                    PZG_LV(A0, j) = i < nd;
                    PZG_LV(A1, j) = i + 1u < nd;
                PZG_LANES_END
                const uint64_t a0 = morel & lanes_ballot(A0), a1 = morel & lanes_ballot(A1);
                if (a0 == 0ull) break;
                LaneVec<uint32_t> Y0, Y1, D0, D1;
                PZG_LANES_BEGIN(j)
                    const uint32_t l4 = PZG_LV(NL, j) - 4u, o0 = 4u * i < l4 ? 4u * i : l4, o1 = 4u * i + 4u < l4 ? 4u * i + 4u : l4;
                    PZG_LV(Y0, j) = lit_load32(mask_select(a0, j, PZG_LV(LAD, j) + o0, lit_a));
                    PZG_LV(Y1, j) = lit_load32(mask_select(a1, j, PZG_LV(LAD, j) + o1, lit_a));
                    PZG_LV(D0, j) = PZG_LV(DL, j) + o0;
                    PZG_LV(D1, j) = PZG_LV(DL, j) + o1;
                PZG_LANES_END
                lds_put_dword(D0, Y0, a0);
                lds_put_dword(D1, Y1, a1);
            }
        }
        if (__builtin_expect(slowl != 0ull, 0)) {  // the run that wraps around the ring's end: byte by byte
            LaneVec<bool> A0;
            for (uint32_t i = 0;; ++i) {
                PZG_LANES_BEGIN(j)
                    PZG_LV(A0, j) = i < PZG_LV(NL, j);
                PZG_LANES_END
                const uint64_t a0 = slowl & lanes_ballot(A0);
                if (a0 == 0ull) break;
                PZG_LANES_BEGIN(j)
                    const uint8_t b = lit_load8(mask_select(a0, j, PZG_LV(LAD, j) + i, lit_a));
                    ring_store(lane_bit(a0, j), (PZG_LV(DL, j) + i) & RMASK, b, j);
                PZG_LANES_END
            }
        }
        PZG_SEQ_ACC(15, tq);
        PZG_MARK("g.far");
        // ---- 2b. the far matches
        if (HYBRID && farm != 0ull) {
            LaneVec<bool> FL4, FL3;
            PZG_LANES_BEGIN(j)
                PZG_LV(FL4, j) = PZG_LV(c.LEN, j) >= 4u;
                PZG_LV(FL3, j) = PZG_LV(c.LEN, j) == 3u;
            PZG_LANES_END
            const uint64_t far4 = farn & lanes_ballot(FL4), far3 = farn & lanes_ballot(FL3);
            for (uint32_t i0 = 0;;) {
                LaneVec<uint32_t> FD[4];
                LaneVec<bool> FA[4];
                PZG_LANES_BEGIN(j)
                    const uint32_t len = PZG_LV(c.LEN, j), l4 = len >= 4u ? len - 4u : 0u, dm = PZG_LV(c.DM, j);
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        PZG_LV(FD[u], j) = dm + (4u * (i0 + u) < l4 ? 4u * (i0 + u) : l4);
                        PZG_LV(FA[u], j) = i0 + u < PZG_LV(FND, j);
                    }
                PZG_LANES_END
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) lds_put_dword(FD[u], FX[u], far4 & lanes_ballot(FA[u]));
                if (i0 == 0u) lds_put_bytes(c.DM, FX[0], far3, far3, far3, 0ull);  // (a three-byte match is one load)
                i0 += 4u;
                PZG_LANES_BEGIN(j)
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) PZG_LV(FA[u], j) = i0 + u < PZG_LV(FND, j);
                PZG_LANES_END
                if (__builtin_expect(lanes_ballot(FA[0]) == 0ull, 1)) break;
                PZG_LANES_BEGIN(j)  // (matches of more than 16 bytes: the next four dwords)
                    const uint32_t len = PZG_LV(c.LEN, j), l4 = len >= 4u ? len - 4u : 0u;
#pragma unroll
                    for (uint32_t u = 0; u < 4u; ++u) {
                        const uint32_t off = 4u * (i0 + u) < l4 ? 4u * (i0 + u) : l4;
#if PZG_DEVICE_PASS
                        PZG_LV(FX[u], j) = far_load32(PZG_LV(FA[u], j) ? PZG_LV(FO, j) + off : FAR_DUMMY);
#else
                        PZG_LV(FX[u], j) = PZG_LV(FA[u], j) ? far_load32(PZG_LV(FO, j) + off) : 0u;
#endif
                    }
                PZG_LANES_END
            }
            if (__builtin_expect(fars != 0ull, 0)) {  // a far match whose output wraps around the ring's end
                LaneVec<bool> A0;
                for (uint32_t i = 0;; ++i) {
                    PZG_LANES_BEGIN(j)
                        PZG_LV(A0, j) = i < PZG_LV(c.LEN, j);
                    PZG_LANES_END
                    const uint64_t a0 = fars & lanes_ballot(A0);
                    if (a0 == 0ull) break;
                    PZG_LANES_BEGIN(j)
                        const uint32_t fb = RES_HIST ? (PZG_LV(FO, j) + i) & HIST_MASK : PZG_LV(FO, j) + i;
#if PZG_DEVICE_PASS
                        const uint8_t b = far_load8(mask_select(a0, j, fb, FAR_DUMMY));
#else
                        const uint8_t b = lane_bit(a0, j) ? far_load8(fb) : (uint8_t)0;
#endif
                        ring_store(lane_bit(a0, j), (PZG_LV(c.DM, j) + i) & RMASK, b, j);
                    PZG_LANES_END
                }
            }
        }
        PZG_SEQ_ACC(7, tq);
        PZG_MARK("g.rounds");
        // ---- 3. matches that read this group's own bytes: the first one still waiting can always go, and with it every one
        // whose source ends in front of it
        while (pend != 0ull) {
            const uint32_t first = ctz64(pend);
            const uint32_t h = lane_get(MO, first);
            if (__builtin_expect((coopm >> first) & 1ull, 0)) {  // a long match: everything in front of it is in place
                seq_coop(op32, h, lane_get(DIST, first), lane_get(LEN, first), run);
                pend &= ~(1ull << first);
                PZG_STAT(24, 1);
                continue;
            }
            LaneVec<bool> RDY;
            PZG_LANES_BEGIN(j)
                PZG_LV(RDY, j) = (int32_t)PZG_LV(SEND, j) <= (int32_t)h;
            PZG_LANES_END
            const uint64_t rdy = pend & lanes_ballot(RDY) & ~coopm;
            seq_copy(c, rdy);
            pend &= ~rdy;
            PZG_STAT(23, 1);
        }
        PZG_MARK("g.end");
        if (RES) {
            // The reference hands out one 32 KiB chunk when its window holds 64 KiB or more, looked at after every MATCH
            // (account_tokens): in a group that is the first match that ends at or behind the mark -- at most one: the window
            // then holds less again for the rest of the group's <= SEQ_GLIM bytes ...
            // (... unless the window held 96 KiB or more when the group began -- 32 KiB or more of literals inside one
            // block, which add to it without a look -- : then the matches that follow hand out a chunk EACH, one after the other, until it
            // holds less than 64 KiB again: counted match by match, as account_tokens() does)
            uint64_t later = hasm;  // the matches that have not been looked at yet
            uint32_t fired = 0u;
            for (;;) {
                LaneVec<bool> FIRE;
                PZG_LANES_BEGIN(j)
                    PZG_LV(FIRE, j) = ow + (PZG_LV(ENDX, j) & 0xffffu) >= 65536u + 32768u * fired;
                PZG_LANES_END
                const uint64_t fire = lanes_ballot(FIRE) & later;
                if (fire == 0ull) break;
                fired += 1u;
                const uint32_t at = ctz64(fire);
                later = at >= 63u ? 0ull : later & (~0ull << (at + 1u));  // the next chunk is the next match's to hand out
            }
            ow += run;
            ow -= 32768u * fired;
            chunks += fired;
        }
        op += run;
        PZG_SEQ_ACC(6, tq);
        return SQ_OK;
    }
    // A match longer than SEQ_CAP at offset m of the group, by all lanes together (copy_match()'s general path, with the group's
    // notion of what is still in the ring: the group's literals and earlier matches have overwritten what is older than
    // op + run - RING).  Every source byte lies in front of the match: for dist < len the pattern repeats with period dist.
    PZG_FN void seq_coop(uint32_t op32, uint32_t m, uint32_t dist, uint32_t len, uint32_t run)
    {
        uint32_t lane = lane_id();
#if PZG_DEVICE_PASS
        asm volatile("" : "+v"(lane));  // (opaque: or the chunks' lane + 64 c + 0.5 are computed once in front of the groups' loop and kept -- spilled -- for its whole life)
#endif
        const uint32_t dst0 = op32 + m, src0 = dst0 - dist;
        const uint32_t fdelta = (uint32_t)(op - flushed);
        constexpr uint32_t MAXCH = (258u + PZG_WAVE - 1u) / PZG_WAVE;
        uint8_t v[MAXCH];
        const bool overlap = dist < len;
#if PZG_DEVICE_PASS
        const float rd = overlap ? __builtin_amdgcn_rcpf((float)dist) : 0.0f;
#endif
#pragma unroll
        for (uint32_t c = 0; c < MAXCH; ++c) {
            const uint32_t k = c * PZG_WAVE + lane;
            if (c * PZG_WAVE < len) {  // wave-uniform
                uint32_t off = k;
                if (overlap) {
#if PZG_DEVICE_PASS
                    uint32_t q = (uint32_t)(((float)k + 0.5f) * rd);
                    off = k - q * dist;
                    off = off >= dist ? off - dist : off;
#else
                    off = k % dist;
#endif
                }
                off = k < len ? off : 0u;
                v[c] = L.ring[(src0 + off) & RMASK];
                if (HYBRID) {
                    const bool far = (k < len) & ((int32_t)(m + off - dist) < (int32_t)(run - RING));
                    if (ballot(far) & far_okmask) {
                        far_fence();
#if PZG_DEVICE_PASS
                        const uint8_t fb = far_load8(far ? far_off(op32, fdelta, m + off - dist) : FAR_DUMMY);
#else
                        const uint8_t fb = far ? far_load8(far_off(op32, fdelta, m + off - dist)) : (uint8_t)0;
#endif
                        v[c] = far ? fb : v[c];
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < MAXCH; ++c) {
            const uint32_t k = c * PZG_WAVE + lane;
            if (c * PZG_WAVE < len) ring_store(k < len, (dst0 + k) & RMASK, v[c], lane);
        }
    }
    PZG_FN uint32_t seq_hot()
    {
        uint32_t why;
        for (;;) {
            why = seq_group();
            if (why != SQ_OK) break;
        }
#if PZG_DEVICE_PASS
        asm volatile("" : "+s"(why));
#endif
        return why;
    }
    // The record at the head of the span on its own: its literals by all lanes (64 bytes a step), its match by copy_match().
    PZG_FN int seq_solo()
    {
        PZG_STAT(24, 1);
        maybe_flush();
        const uint32_t rec = lane_get(QTN, 0u), nl = seq_nl(rec), len = seq_len(rec), dist = seq_dist(rec);
        const uint32_t lit_a = reg_lit(lane_get(QINFO, 0u) & 255u) + s_lc;
        const uint32_t lane = lane_id();
#pragma nounroll
        for (uint32_t k0 = 0; k0 < nl; k0 += PZG_WAVE) {
            const uint32_t i = k0 + lane;
            const uint8_t b = lit_load8(lit_a + (i < nl ? i : nl - 1u));
            ring_store(i < nl, ((uint32_t)op + i) & RMASK, b, lane);
        }
        op += nl;
        if (RES) ow += nl;  // (no match: no moveWindow check)
        maybe_flush();
        if (len != 0u) {
            if ((uint64_t)dist > op + (RING_BITS == 15 ? hist_extra : 0u)) return fail(ST_BAD_DISTANCE, dist, (uint32_t)op);
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
            prof[14] += 1;
#endif
            copy_match(dist, len);
            maybe_flush();
            if (RES) account_tokens(len, 1u, true);
        }
        s_lc += nl;
        s_rd += 1u;
        if (seq_refill()) s_lc = 0u;
        return ST_OK;
    }

    template <bool FX>
    PZG_FN int token_loop()
    {
        bool strips = STRIPS && strip != nullptr, tight = false;
        for (;;) {
            PZG_T0(tw);
            bool checked;
            bool span_done = false;
            if (strips) {
                bool poor = false;
                const int ss = strip_span<FX>(checked, poor, tight);
                if (ss != ST_OK && ss != STRIP_NA) return ss;  // (an error first: a poor span can end in one too)
                if (ss == STRIP_NA || poor) strips = false;
                if (ss == ST_OK && !checked) continue;
                if (ss == ST_OK) span_done = true;
            }
            if (!span_done) {
            uint32_t why = HL_GENERAL;
            LaneVec<uint32_t> TK0, TK1;
            uint64_t S0 = 0, S1 = 0;
            uint32_t k0 = 0, k1 = 0;
            if (!RES) why = FX ? hot_loop<FX, 0>(TK0, TK1, S0, S1, k0, k1) : use_sub ? hot_loop<FX, 1>(TK0, TK1, S0, S1, k0, k1)
                                                                                       : hot_loop<FX, 0>(TK0, TK1, S0, S1, k0, k1);
            PZG_T0(trw);
            if (why == HL_WINDOW) checked = window2_rare(TK0, TK1, S0, S1, k0, k1);
            else checked = qn < QHIGH && fill_queue<FX>();
            PZG_HOT_ACC(10, trw);
            PZG_ACC(4, tw);
            if (!checked) {
                PZG_T0(te);
                const int se = emit_segment();
                PZG_ACC(12, te);
                if (se) return se;
                continue;
            }
            }
            PZG_T0(tc);
            const int st = token_step_checked<FX>();
            PZG_STAT(12, 1);  // checked steps
            PZG_ACC(5, tc);
#if defined(PZG_PROFILE) && PZG_DEVICE_PASS
            prof[15] += 1;
#endif
            if (st == ST_OK) continue;
            // end of block or error: first everything that precedes it in the stream (an error there wins)
            while (qn != 0u) {
                const int se = emit_segment();
                if (se) return se;
            }
            complete_pending();
            return st == STEP_EOB ? ST_OK : st;
        }
    }

    // ---- Deflate.hs:70-78: stored block -------------------------------------------------------------
    PZG_FN int stored_block()
    {
        br.align_to_byte();  // advanceToByte (Monad.hs:304-307)
        if (br.avail() < 32) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t w = br.peek32();
        const uint32_t len = w & 0xffffu, nlen = w >> 16;
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
#pragma nounroll
            for (uint32_t k0 = 0; k0 < piece; k0 += PZG_WAVE) {
                const uint32_t k = k0 + lane;
                const uint8_t v = in[p + done + (k < piece ? k : piece - 1u)];
                ring_store(k < piece, (((uint32_t)op + k)) & RMASK, v, lane);
            }
            op += piece;
            done += piece;
        }
        maybe_flush();
        in_byte0 = p + len;
        br.start(in, in_len, in_byte0);
        return ST_OK;
    }

    // ---- Deflate.hs:79-82,241-251: the fixed code ---------------------------------------------------
    static constexpr uint32_t FIXED_MAGIC = 0x51DF1BEDu;
    PZG_FN void load_fixed_tables()
    {
        lit_n = 288u;
        dist_n = 32u;
        if (uni(L.fixed_ready) == FIXED_MAGIC) {  // still there from an earlier block or stream of this wave
            lit_e15 = dist_e15 = 32768u;
            use_sub = FX_TABLES ? 0u : 1u;  // as it was left when the tables were built (see below)
            // (lit_sub_used / dist_sub_used are what that build left too: run() zeroes them -- the fixed code's own table shape has no
            // second level -- and a resumable decoder, whose LDS image is its own, carries them in its ResumeState)
            return;
        }
        const uint32_t lane = lane_id();
#pragma nounroll
        for (uint32_t s0 = 0; s0 < 320u; s0 += PZG_WAVE) {
            const uint32_t s = s0 + lane;
            if (s < 320u) L.lens[s] = (uint8_t)(s <= 143u ? 8u : s <= 255u ? 9u : s <= 279u ? 7u : s <= 287u ? 8u : 5u);
        }
        if (FX_TABLES) {  // one 2^9-entry literal/length table over both primary tables, a 2^5-entry distance table in the pool
            build_table<9, TREE_LITLEN>(L.lens, 288u, L.lit_lut, &L.lit_meta, &lit_e15);
            build_table<5, TREE_DIST>(L.lens + 288u, 32u, L.sub, &L.dist_meta, &dist_e15);
            use_sub = 0u;
        } else {
            // (the resumable instance keeps the dynamic blocks' shape: use_sub = 1 from build_table(), the fixed code's 56
            // long prefixes -- literals 144..255 are 9 bits -- go through the second-level tables)
            build_table<LIT_BITS, TREE_LITLEN>(L.lens, 288u, L.lit_lut, &L.lit_meta, &lit_e15);
            build_table<DIST_BITS, TREE_DIST>(L.lens + 288u, 32u, L.dist_lut, &L.dist_meta, &dist_e15);
        }
        if (lane == 0u || PZG_WAVE == 1u) L.fixed_ready = FIXED_MAGIC;
        wave_sync();
    }

    // ---- Deflate.hs:83-101,124-156: dynamic block header ---------------------------------------------
    // ---- code lengths, 64 bit offsets at a time (getCodeLengths, Deflate.hs:124-156, wave-parallel) -------------------
    // Lane k decodes the code-length symbol that would start k bits ahead of the cursor; the scalar walk finds the real
    // ones; they are ranked onto lanes 0.. in stream order, their run lengths prefix-summed into positions in `lens`, a
    // "repeat previous" (16) takes the value of the nearest symbol before it that defines one, and every symbol's run is
    // stored.  It stops in front of the first entry that is no symbol (dynamic_header()'s serial step reports it) and
    // right after the symbol that completes HLIT + HDIST lengths -- exactly where the serial loop stops.
    // Precondition: br.window_ok() (no symbol of the window reaches past the stream), n < maxl.
    // Returns false when the entry at the cursor is one for the serial step.
    PZG_FN bool cl_window(uint32_t &n, uint32_t &prev, uint32_t maxl)
    {
        const uint32_t i0 = br.chunk0 + (br.rp >> 5), boff = br.rp & 31u;
        const uint32_t B0 = br.dword(i0), B1 = br.dword(i0 + 1u), B2 = br.dword(i0 + 2u), B3 = br.dword(i0 + 3u);
        LaneVec<uint32_t> TB, INFO;  // INFO: run length | value << 8 | defines a value << 12 | bits << 16 | offset << 24
        PZG_LANES_BEGIN(k)
            const uint32_t q = boff + k, sel = q >> 5, r = q & 31u;
            const uint32_t lo = sel == 0u ? B0 : sel == 1u ? B1 : B2, hi = sel == 0u ? B1 : sel == 1u ? B2 : B3;
            const uint32_t w = funnel(hi, lo, r);
            const uint32_t e = L.dist_lut[w & ((1u << CL_BITS) - 1u)];
            const uint32_t sym = ent_val(e), cn = ent_n(e), ce = ent_cl_extra(e);
            const uint32_t extra = (w >> cn) & ((1u << ce) - 1u);
            const uint32_t tb = cn + ce;  // <= 7 + 7
            const uint32_t num = sym <= 15u ? 1u : sym == 18u ? 11u + extra : 3u + extra;
            PZG_LV(TB, k) = ent_is_stop(e) ? 128u : tb;
            PZG_LV(INFO, k) = (num & 0xffu) | ((sym <= 15u ? sym : 0u) << 8) | ((sym != 16u ? 1u : 0u) << 12) | ((tb & 15u) << 16) | (k << 24);
        PZG_LANES_END
        uint64_t S = 0;
        const uint32_t kend = walk_half(TB, 0u, S);
        uint32_t consumed = kend + 64u;
        if (kend >= 64u) {  // ran into an entry that is no symbol: it stays at the cursor
            consumed = 63u - clz64(S);
            S &= ~(1ull << consumed);
        }
        if (S == ~0ull) {  // (64 one-bit symbols: lane 63 is where the ranking sends what it discards)
            S &= ~(1ull << 63);
            consumed = 63u;
        }
        const uint32_t cnt = popc64(S);
        if (cnt == 0u) return false;
        LaneVec<uint32_t> DEST, D, INCL, DEFI;
        PZG_LANES_BEGIN(k)
            PZG_LV(DEST, k) = mask_select(S, k, mbcnt_k(S, k), 63u);
        PZG_LANES_END
        lanes_scatter(D, INFO, DEST);
        PZG_LANES_BEGIN(j)
            const uint32_t d = PZG_LV(D, j);
            PZG_LV(INCL, j) = j < cnt ? (d & 0xffu) : 0u;
            PZG_LV(DEFI, j) = (j < cnt && ((d >> 12) & 1u) != 0u) ? j + 1u : 0u;
        PZG_LANES_END
        lanes_iscan_add(INCL);  // lengths written up to and including symbol j
        lanes_iscan_max(DEFI);  // 1 + the nearest symbol at or before j that defines a value (0: none in this window)
        LaneVec<bool> FULL;
        PZG_LANES_BEGIN(j)
            PZG_LV(FULL, j) = (j < cnt) & (n + PZG_LV(INCL, j) >= maxl);
        PZG_LANES_END
        const uint64_t full = lanes_ballot(FULL);
        uint32_t use = cnt;
        if (full != 0ull) {  // the serial loop reads no symbol past the one that completes maxl lengths
            use = ctz64(full) + 1u;
            const uint32_t dl = lane_get(D, use - 1u);
            consumed = (dl >> 24) + ((dl >> 16) & 15u);
        }
        LaneVec<uint32_t> SRC, DV, VAL, NUMX;
        PZG_LANES_BEGIN(j)
            const uint32_t s1 = PZG_LV(DEFI, j);
            PZG_LV(SRC, j) = s1 != 0u ? s1 - 1u : 0u;
        PZG_LANES_END
        lanes_gather(DV, D, SRC);
        PZG_LANES_BEGIN(j)
            const uint32_t val = PZG_LV(DEFI, j) != 0u ? ((PZG_LV(DV, j) >> 8) & 15u) : prev;
            PZG_LV(VAL, j) = val;
            // `lens` is all zeros when the header starts (dynamic_header): only runs of a length that is not 0 are stored --
            // at most 6 long (symbol 16), where a run of zeros may be 138
            PZG_LV(NUMX, j) = (j < use && val != 0u) ? (PZG_LV(D, j) & 0xffu) : 0u;
        PZG_LANES_END
        lanes_iscan_max(NUMX);
        const uint32_t maxnum = lane_get(NUMX, 63u);  // (wave-uniform trip count: no lane-dependent loop exit)
#pragma nounroll
        for (uint32_t r = 0; r < maxnum; ++r) {
            PZG_LANES_BEGIN(j)
                const uint32_t num = PZG_LV(D, j) & 0xffu;
                const bool on = (j < use) & (r < num) & (PZG_LV(VAL, j) != 0u);
                const uint32_t pos = on ? n + PZG_LV(INCL, j) - num + r : 0u;
                sel_store(on, &L.lens[pos], (uint8_t)PZG_LV(VAL, j), j);
            PZG_LANES_END
        }
        prev = lane_get(VAL, use - 1u);  // (16 leaves it as it was, 17 / 18 make it 0, a length makes it that length)
        n += lane_get(INCL, use - 1u);
        br.drop(consumed);
        return true;
    }

    PZG_FN int dynamic_header(uint32_t block_bit)
    {
        const uint32_t lane = lane_id();
        PZG_T0(th0);
        if (br.avail() < 14) return fail(ST_TRUNCATED, 0, 0);
        uint32_t w = br.peek32();
        const uint32_t hlit = 257u + (w & 31u);
        const uint32_t hdist = 1u + ((w >> 5) & 31u);
        const uint32_t hclen = 4u + ((w >> 10) & 15u);
        br.drop(14);
        // hclen x 3-bit lengths in codeLengthOrder (Deflate.hs:87-88,290-292)
        for (uint32_t i0 = 0; i0 < 20u; i0 += PZG_WAVE) sel_store(i0 + lane < 20u, &L.cl_lens[(i0 + lane) % 20u], 0, lane);
        if (!RES) {  // cl_window() stores no zeros
            constexpr uint32_t NDW = (uint32_t)(sizeof(L.lens) / 4u);
            static_assert(sizeof(L.lens) % 4u == 0u, "lens is cleared by dwords");
            uint32_t *lz = reinterpret_cast<uint32_t *>(L.lens);
#pragma nounroll
            for (uint32_t i0 = 0; i0 < NDW; i0 += PZG_WAVE) {
                const uint32_t i = i0 + lane;
                if (PZG_WAVE == 1u) { for (uint32_t t = 0; t < NDW; ++t) lz[t] = 0u; break; }
                lz[i < NDW ? i : NDW - 1u] = 0u;
            }
        }
        wave_sync();
        if (br.avail() < (int64_t)(3u * hclen)) return fail(ST_TRUNCATED, 0, 0);
        // codeLengthOrder = 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 packed 5 bits each
        const uint64_t ORD_LO = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 |
                                6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
        const uint64_t ORD_HI = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
#pragma nounroll
        for (uint32_t i0 = 0; i0 < hclen; i0 += 10u) {  // up to 10 fields (30 bits) per peek
            w = br.peek32();
            const uint32_t m = hclen - i0 < 10u ? hclen - i0 : 10u;
#pragma nounroll
            for (uint32_t j0 = 0; j0 < m; j0 += PZG_WAVE) {
                const uint32_t j = j0 + lane;
                const uint32_t i = i0 + j;
                const uint32_t sym = i < 12u ? (uint32_t)(ORD_LO >> (5u * (i % 12u))) & 31u
                                             : (uint32_t)(ORD_HI >> (5u * ((i - 12u) % 7u))) & 31u;
                sel_store(j < m, &L.cl_lens[sym % 20u], (uint8_t)((w >> (3u * (j % 10u))) & 7u), lane);
            }
            br.drop(3u * m);
        }
        if (lane == 0u || PZG_WAVE == 1u) L.fixed_ready = 0u;  // the tables are about to be overwritten
        lit_n = hlit;
        uint32_t cl_e15;
        if (!build_table<CL_BITS, TREE_CODELEN>(L.cl_lens, 19u, L.dist_lut, &L.lit_meta, &cl_e15)  /* (its meta is never read; lit_meta is rebuilt below) */)
            return fail(ST_HUFF_BUILD, TREE_CODELEN, block_bit);
        PZG_HACC(8, th0);
        PZG_T0(th1);
        // getCodeLengths (Deflate.hs:124-156) over HLIT+HDIST as ONE sequence
        const uint32_t maxl = hlit + hdist;
        uint32_t n = 0, prev = 0;
        while (n < maxl) {
            if (!RES && br.window_ok() && cl_window(n, prev, maxl)) continue;  // (the resumable instance stays serial: it may suspend here)
            w = br.peek32();
            const uint32_t e = uni(L.dist_lut[w & ((1u << CL_BITS) - 1u)]);
            if (int st = check_entry(e)) return st;
            const uint32_t sym = ent_val(e), cn = ent_n(e), ce = ent_cl_extra(e);
            if (br.avail() < (int64_t)(cn + ce)) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t extra = (w >> cn) & ((1u << ce) - 1u);
            br.drop(cn + ce);
            if (sym <= 15u) {
                sel_store(lane == 0u, &L.lens[n], (uint8_t)sym, lane);
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
#pragma nounroll
            for (uint32_t k0 = 0; k0 < num; k0 += PZG_WAVE)
                sel_store(k0 + lane < num, &L.lens[n + k0 + lane], (uint8_t)val, lane);
            n += num;
        }
        wave_sync();
        PZG_HACC(9, th1);
        PZG_T0(th2);
        // litTree first, then distTree (Deflate.hs:98-99): errors surface in that order
        if (!build_table<LIT_BITS, TREE_LITLEN>(L.lens, hlit, L.lit_lut, &L.lit_meta, &lit_e15))
            return fail(ST_HUFF_BUILD, TREE_LITLEN, block_bit);
        PZG_HACC(10, th2);
        PZG_T0(th3);
        dist_n = n - hlit;
        if (!build_table<DIST_BITS, TREE_DIST>(L.lens + hlit, n - hlit, L.dist_lut, &L.dist_meta, &dist_e15))
            return fail(ST_HUFF_BUILD, TREE_DIST, block_bit);
        PZG_HACC(11, th3);
        return ST_OK;
    }

    // ---- Zlib.hs:53-69 inflateWithHeaders + Deflate.hs:39-63 inflate ---------------------------------
    PZG_FN void run(const uint8_t *in_, uint64_t in_len_, uint8_t *out_, uint64_t cap_, StreamResult *res, const uint8_t *dict_ = nullptr,
                    uint32_t dict_len_ = 0)
    {
        hist = nullptr;
        dict = dict_;
        dict_len = dict_len_;
        in = in_;
        in_len = in_len_;
        out = out_;
        cap = cap_;
        op = 0;
        flushed = 0;
        adler_a = 1;
        adler_b = 0;
        lit_e15 = dist_e15 = 0;
        lit_n = dist_n = 0;
        use_sub = 0;
        lit_sub_used = 0;
        dist_sub_used = 0;
        s_rd = s_qn = s_total = s_cnt = s_lc = 0;
        pend_m0 = pend_m1 = 0;
        pend_pos = 0;
        qn = 0;
        status = ST_OK;
        detail0 = detail1 = 0;
        in_byte0 = 0;
        hist_extra = 0;
        gz_expect = 0;
        res_final = 1;
        phase = bfinal_cur = stored_left = deferred = ow = chunks = 0;
        susp_pos = 0;
        in_total_bits = 0;
#if PZG_DEVICE_PASS && PZG_DMA_PREFETCH
        br.pf = L.pf;
#endif
        set_far_base();
        br.start(in, in_len, 0);
        PZG_T0(tall);
        // the reader indexes dwords with 32 bits: 16 GiB per stream (include/pzg.h).  (Tested on the high word: a 64-bit
        // compare is a vector instruction whose constant would sit in a vector-register pair for the whole kernel.)
        uint32_t len_hi = (uint32_t)(in_len >> 32);
#if PZG_DEVICE_PASS
        asm volatile("" : "+s"(len_hi));  // (opaque: or the optimizer folds the test back into the 64-bit compare)
#endif
        if (len_hi >> 2) fail(ST_TRUNCATED, 0, 0);
        else decode();
        PZG_ACC(0, tall);
#if PZG_DEVICE_PASS && PZG_DMA_PREFETCH
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no fetch into this wave's LDS may outlive the stream
#endif
        uint64_t used_bits = stream_bit_pos();
        uint64_t used = (used_bits + 7u) >> 3;
        if (used > in_len) used = in_len;
        if (HYBRID && op > cap && (status == ST_OK || status == ST_CHECKSUM || status == ST_GZIP_ISIZE)) {
            // bytes past the capacity were never stored, so far reads of them (and the checksum) are not
            // to be trusted: the 32 KiB-ring kernel, which needs nothing but LDS, redoes this stream
            status = ST_RETRY_FULL_RING;
        }
        if ((status == ST_OK || (GZIP && status == ST_GZIP_ISIZE)) && op > cap) status = ST_OUT_TOO_SMALL;  // (gzip: nothing stored to check the CRC of)
        res->status = status;
        res->detail0 = detail0;
        res->detail1 = detail1;
        // (a stream that FAILED after it had outgrown its capacity: what lies past the capacity was never stored, on the small rings
        // the far reads of it were skipped, so the running checksum means nothing -- and must not depend on what the wave decoded before)
        const bool unstored = op > cap && status != ST_OK && status != ST_OUT_TOO_SMALL && status != ST_RETRY_FULL_RING;
        res->adler = unstored ? 0u : (adler_b << 16) | adler_a;
        res->gz_crc = gz_expect;
        res->out_len = op;
        res->in_used = used;
    }

    // ---- RFC 1952 (extension; the reference lists gzip as a TODO, README.md:42-50) -------------------
    PZG_FN int gz_byte(uint32_t &b, uint32_t &hreg)
    {
        if (br.avail() < 8) return fail(ST_TRUNCATED, 0, 0);
        b = br.peek32() & 0xffu;
        br.drop(8);
        uint32_t r = hreg ^ b;  // CRC-32 of the header bytes, for FHCRC
#pragma nounroll
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (0xedb88320u & (0u - (r & 1u)));
        hreg = r;
        return ST_OK;
    }
    PZG_FN int gzip_header()
    {
        uint32_t hreg = 0xffffffffu, id1 = 0, id2 = 0, cm = 0, flg = 0, b = 0;
        if (int st = gz_byte(id1, hreg)) return st;
        if (int st = gz_byte(id2, hreg)) return st;
        if (id1 != 0x1fu || id2 != 0x8bu) return fail(ST_GZIP_HEADER, 1, (id1 << 8) | id2);
        if (int st = gz_byte(cm, hreg)) return st;
        if (cm != 8u) return fail(ST_GZIP_HEADER, 2, cm);
        if (int st = gz_byte(flg, hreg)) return st;
        if (flg & 0xe0u) return fail(ST_GZIP_HEADER, 3, flg);
#pragma nounroll
        for (int k = 0; k < 6; ++k)  // MTIME, XFL, OS
            if (int st = gz_byte(b, hreg)) return st;
        if (flg & 4u) {  // FEXTRA
            uint32_t lo = 0, hi = 0;
            if (int st = gz_byte(lo, hreg)) return st;
            if (int st = gz_byte(hi, hreg)) return st;
#pragma nounroll
            for (uint32_t k = 0; k < (lo | (hi << 8)); ++k)
                if (int st = gz_byte(b, hreg)) return st;
        }
#pragma nounroll
        for (uint32_t field = 8u; field <= 16u; field <<= 1)  // FNAME, FCOMMENT: zero-terminated
            if (flg & field) {
                do {
                    if (int st = gz_byte(b, hreg)) return st;
                } while (b != 0u);
            }
        if (flg & 2u) {  // FHCRC: the low 16 bits of the CRC-32 of the header so far
            const uint32_t want = ~hreg & 0xffffu;
            uint32_t lo = 0, hi = 0, dummy = 0;
            if (int st = gz_byte(lo, dummy)) return st;
            if (int st = gz_byte(hi, dummy)) return st;
            if ((lo | (hi << 8)) != want) return fail(ST_GZIP_HEADER, 4, lo | (hi << 8));
        }
        return ST_OK;
    }

    // Adler-32 of the preset dictionary (PZG_FDICT), 64 bytes per step: A += sum d ; B += 64 A_before + sum (64 - j) d_j
    PZG_FN uint32_t dict_adler()
    {
        uint32_t a = 1, bsum = 0;
        const uint32_t lane = lane_id();
#pragma nounroll
        for (uint32_t k0 = 0; k0 < dict_len; k0 += 64u) {
            const uint32_t n = dict_len - k0 < 64u ? dict_len - k0 : 64u;
            uint32_t sa = 0, sb = 0;
#if PZG_DEVICE_PASS
            const uint32_t v = lane < n ? dict[k0 + lane] : 0u;
            sa = wave_sum(v);
            sb = wave_sum((n - lane) * v);  // (lanes >= n hold 0)
#else
            for (uint32_t j = 0; j < n; ++j) {
                sa += dict[k0 + j];
                sb += (n - j) * dict[k0 + j];
            }
            (void)lane;
#endif
            bsum = (uint32_t)(((uint64_t)bsum + (uint64_t)n * a + sb) % ADLER_MOD);
            a = (a + sa) % ADLER_MOD;
        }
        return (bsum << 16) | a;
    }
    // the last min(dict_len, 32768) dictionary bytes become the history in front of the output (ring positions -1, -2, ...)
    PZG_FN void install_dictionary()
    {
        const uint32_t lane = lane_id();
        const uint32_t use = dict_len < 32768u ? dict_len : 32768u;
#pragma nounroll
        for (uint32_t k0 = 0; k0 < use; k0 += PZG_WAVE) {
            const uint32_t k = k0 + lane;  // byte `k + 1` positions before the output start
            const uint8_t v = dict[dict_len - 1u - (k < use ? k : use - 1u)];
            ring_store(k < use, ((0u - 1u - k)) & RMASK, v, lane);
        }
        hist_extra = use;
        wave_sync();
    }

    // Zlib.hs:55-68: CMF, FLG; FCHECK, then CM, then CINFO; FDICT
    PZG_FN int zlib_header()
    {
        if (br.avail() < 16) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t hw = br.peek32();
        const uint32_t cmf = hw & 0xffu, flg = (hw >> 8) & 0xffu;
        // (resumable instance: running out of input below -- the DICTID of an FDICT header -- suspends at the start of the
        // header, which is then read again as a whole; the three checks on CMF/FLG come first, as in the reference)
        br.drop(16);
        if (((cmf << 8) | flg) % 31u != 0u) return fail(ST_HDR_FCHECK, (cmf << 8) | flg, 0);
        if ((cmf & 15u) != 8u) return fail(ST_HDR_METHOD, cmf & 15u, 0);
        if ((cmf >> 4) > 7u) return fail(ST_HDR_WINDOW, cmf >> 4, 0);
        if (flg & 0x20u) {
            if (br.avail() < 32) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t t = br.peek32();
            br.drop(32);
            if (dict_len != 0u) {  // extension (PZG_FDICT): a dictionary was supplied for this stream
                if (RING_BITS != 15) return fail(ST_RETRY_FULL_RING, 0, 0);  // decoded by the 32 KiB-ring instance only
                const uint32_t theirs = (t << 24) | ((t & 0xff00u) << 8) | ((t >> 8) & 0xff00u) | (t >> 24);
                const uint32_t ours = dict_adler();
                if (theirs != ours) return fail(ST_DICT, theirs, ours);
                install_dictionary();
            }
            // (none supplied -- Zlib.hs:68: skip DICTID, carry on with an empty history)
        }
        return ST_OK;
    }

    PZG_FN int decode()
    {
        if (GZIP) {
            // RFC 1952 2.2: a gzip file is a series of members; they are decoded one after the other into one output.
            // Each member's ISIZE is checked here against what it produced; the CRC-32s are folded into the CRC the
            // whole output must have (crc32_append), which crc32_verify_kernel then checks in one pass.
            uint64_t mstart = 0;
            gz_expect = 0;
            for (;;) {
                if (int st = gzip_header()) return st;
                if (int st = blocks()) return st;
                br.align_to_byte();
                if (br.avail() < 64) return fail(ST_TRUNCATED, 0, 0);
                const uint32_t crc = br.peek32();
                br.drop(32);
                const uint32_t isize = br.peek32();
                br.drop(32);
                const uint64_t mlen = op - mstart;
                gz_expect = crc32_append(gz_expect, crc, mlen);
                // (reported unless the CRC-32 is wrong as well: crc32_verify_kernel looks at that first, as zlib does)
                if (isize != (uint32_t)mlen) {
                    flush_to(op);
                    return fail(ST_GZIP_ISIZE, isize, (uint32_t)mlen);
                }
                mstart = op;
                if (br.avail() < 16 || (br.peek32() & 0xffffu) != 0x8b1fu) break;  // no further member follows
            }
            flush_to(op);  // (once, at the very end: flushes move whole 16-byte groups, so only the last may end on an odd byte)
            return ST_OK;
        }
        if (int st = zlib_header()) return st;
        if (int st = blocks()) return st;
        return zlib_trailer();
    }

    // Deflate.hs:52-63 checkChecksum: align, fold the rest of the window, compare big-endian
    PZG_FN int zlib_trailer()
    {
        flush_to(op);
        br.align_to_byte();
        if (br.avail() < 32) return fail(ST_TRUNCATED, 0, 0);
        const uint32_t t = br.peek32();
        const uint32_t theirs = (t << 24) | ((t & 0xff00u) << 8) | ((t >> 8) & 0xff00u) | (t >> 24);
        br.drop(32);
        const uint32_t ours = (adler_b << 16) | adler_a;
        if (theirs != ours) return fail(ST_CHECKSUM, theirs, ours);
        return ST_OK;
    }

    // Deflate.hs:45-50 go: blocks up to and including the final one
    PZG_FN int blocks()
    {
        for (;;) {
            pin_uniform();
            const uint32_t block_bit = (uint32_t)stream_bit_pos();
            if (br.avail() < 3) return fail(ST_TRUNCATED, 0, 0);
            const uint32_t bh = br.peek32();
            const uint32_t bfinal = bh & 1u, btype = (bh >> 1) & 3u;
            br.drop(3);
            int st;
            if (btype == 0u) {
                st = stored_block();
            } else if (btype == 3u) {
                st = fail(ST_FMT_BTYPE, 3, 0);
            } else {
                PZG_T0(th);
                if (btype == 1u) {
                    load_fixed_tables();
                    st = ST_OK;
                } else {
                    st = dynamic_header(block_bit);
                }
                PZG_ACC(1, th);
                PZG_T0(tt);
                if (st == ST_OK) st = (FX_TABLES && btype == 1u) ? token_loop<FX_TABLES>() : token_loop<false>();
                PZG_ACC(2, tt);
            }
            if (st != ST_OK) return st;
            if (bfinal) return ST_OK;
        }
    }

    // ---- the resumable instance: decompressIncremental (Monad.hs:163-197, Zlib.hs decompressIncremental) -----------
    // One call decodes as far as this call's input and output room allow and leaves the decoder in a ResumeState:
    //   ST_NEED_INPUT  every complete element of the input has been consumed (the reference's NeedMore); the next call
    //                  passes the unconsumed tail (from in_used on) followed by new input
    //   ST_OUT_FULL    the output room is used up; call again with the rest of the input and fresh room
    //   ST_OK / error  the stream has ended
    // Running out of input inside an element (a header, a token, the trailer) rewinds to the start of that element:
    // nothing of it has been acted upon, so re-reading it with more input behind it is the same as the reference
    // resuming mid-element.  Stored blocks are copied as far as input and room go.
    PZG_FN int suspend_input(uint64_t at_bit)
    {
        susp_pos = at_bit;
        status = ST_OK;
        detail0 = detail1 = 0;
        return ST_NEED_INPUT;
    }
    PZG_FN bool truncated_suspends(int st) const { return st == ST_TRUNCATED && !res_final; }

    // the tokens of one block: token_loop() with the two ways out that a resumable decoder adds
    PZG_FN int token_loop_res()
    {
        if (deferred != 0u) susp_pos = stream_bit_pos();  // (resumed while draining: nothing of this call's input is consumed yet)
        bool strips = STRIPS && strip != nullptr, tight = false;
        for (;;) {
            int st;
            bool span_stopper = false;
            if (deferred == 0u && strips) {
                // (round 5) a span of strips while the call's input and room have space for one: decoded and emitted inside this call
                bool poor = false;
                const int ss = strip_span<false>(span_stopper, poor, tight);
                if (ss == ST_OUT_FULL) {
                    susp_pos = stream_bit_pos();
                    return ST_OUT_FULL;
                }
                if (ss != ST_OK && ss != STRIP_NA) return ss;
                if (ss == STRIP_NA || poor) strips = false;
                if (ss == ST_OK && !span_stopper) continue;
            }
            if (deferred == 0u) {
                // (round 4: the resumable decoder runs the same hot loop as the batch kernel for as long as whole 128-bit
                // windows lie inside this call's input and nothing special is due; everything else -- the end of the input,
                // a full room, the reference's chunk accounting across a 64 KiB boundary -- stays with the general code)
                LaneVec<uint32_t> TK0, TK1;
                uint64_t S0 = 0, S1 = 0;
                uint32_t k0 = 0, k1 = 0;
                const uint32_t why = (PZG_RES_HOT_LOOP && !span_stopper) ? (use_sub ? hot_loop<false, 1>(TK0, TK1, S0, S1, k0, k1) : hot_loop<false, 0>(TK0, TK1, S0, S1, k0, k1))
                                                                          : (uint32_t)HL_GENERAL;
                const bool checked = span_stopper || (why == HL_WINDOW ? window2_rare(TK0, TK1, S0, S1, k0, k1) : (qn < QHIGH && fill_queue<false>()));
                if (!checked) {
                    const int se = emit_segment();
                    if (se == ST_OUT_FULL) {
                        susp_pos = stream_bit_pos();
                        return ST_OUT_FULL;
                    }
                    if (se) return se;
                    continue;
                }
                const uint64_t tok_pos = stream_bit_pos();  // where the token about to be decoded starts
                st = token_step_checked<false>();
                if (st == ST_OK) continue;
                if (truncated_suspends(st)) {
                    // the token is not all there yet: everything before it is decoded first, then the input is asked for
                    status = ST_OK;
                    susp_pos = tok_pos;
                    st = ST_NEED_INPUT;
                } else {
                    susp_pos = stream_bit_pos();
                }
                deferred = (uint32_t)st;
            }
            // end of block, an error, or the input ran out: first everything that precedes it in the stream
            while (qn != 0u) {
                const int se = emit_segment();
                if (se == ST_OUT_FULL) return ST_OUT_FULL;  // (deferred stays set: the next call goes on draining)
                if (se) return se;
            }
            complete_pending();
            st = (int)deferred;
            deferred = 0u;
            if (st == ST_NEED_INPUT) return ST_NEED_INPUT;
            if (st != STEP_EOB) {  // the error found behind the queued tokens (its detail words were kept by fail())
                status = st;
                return st;
            }
            return ST_OK;
        }
    }

    PZG_FN int resume_decode()
    {
        for (;;) {
            pin_uniform();
            if (phase == PH_HEADER) {
                const int st = zlib_header();
                if (truncated_suspends(st)) return suspend_input(0);
                if (st) return st;
                phase = PH_BLOCK;
            } else if (phase == PH_BLOCK) {
                const uint64_t blk_pos = stream_bit_pos();
                const uint32_t block_bit = (uint32_t)(in_total_bits + blk_pos);  // (from the start of the whole stream)
                if (br.avail() < 3) {
                    if (!res_final) return suspend_input(blk_pos);
                    return fail(ST_TRUNCATED, 0, 0);
                }
                const uint32_t bh = br.peek32();
                const uint32_t btype = (bh >> 1) & 3u;
                bfinal_cur = bh & 1u;
                br.drop(3);
                if (btype == 0u) {
                    br.align_to_byte();  // advanceToByte (Monad.hs:304-307)
                    if (br.avail() < 32) {
                        if (!res_final) return suspend_input(blk_pos);
                        return fail(ST_TRUNCATED, 0, 0);
                    }
                    const uint32_t w = br.peek32();
                    const uint32_t len = w & 0xffffu, nlen = w >> 16;
                    if (len != ((~nlen) & 0xffffu)) return fail(ST_FMT_LEN_NLEN, len, nlen);
                    br.drop(32);
                    stored_left = len;
                    ow += len;  // (emitBlock adds the block to the window as a whole, Deflate.hs:77)
                    phase = PH_STORED;
                } else if (btype == 3u) {
                    return fail(ST_FMT_BTYPE, 3, 0);
                } else {
                    if (btype == 1u) {
                        load_fixed_tables();
                    } else {
                        const int st = dynamic_header(block_bit);
                        if (truncated_suspends(st)) return suspend_input(blk_pos);
                        if (st) return st;
                    }
                    phase = PH_TOKENS;
                }
            } else if (phase == PH_STORED) {
                // Monad.hs:265-293 nextBlock + Monad.hs:317-322 emitBlock, as far as the input and the room go
                const uint64_t p = stream_bit_pos() >> 3;  // byte offset of the raw data in this call's input
                uint64_t n = stored_left;
                bool more_input = false, more_room = false;
                if (p + n > in_len) {
                    n = in_len - p;
                    more_input = true;
                }
                if (op + n + 512u > cap) {
                    n = cap > op + 512u ? (n < cap - op - 512u ? n : cap - op - 512u) : 0u;
                    more_room = true;
                    more_input = false;
                }
                stored_copy(p, (uint32_t)n);
                stored_left -= (uint32_t)n;
                if (stored_left != 0u) {
                    susp_pos = stream_bit_pos();
                    if (more_room) return ST_OUT_FULL;
                    if (more_input && !res_final) return suspend_input(susp_pos);
                    return fail(ST_TRUNCATED, 0, 0);
                }
                move_window_check();  // Deflate.hs:47
                phase = bfinal_cur ? PH_TRAILER : PH_BLOCK;
            } else if (phase == PH_TOKENS) {
                const int st = token_loop_res();
                if (st) return st;
                move_window_check();  // Deflate.hs:47
                phase = bfinal_cur ? PH_TRAILER : PH_BLOCK;
            } else if (phase == PH_TRAILER) {
                const uint64_t tpos = stream_bit_pos();
                // (asked for before zlib_trailer() flushes: a flush to an odd position must be the last one)
                if (br.avail() < (int64_t)(((8u - (br.rp & 7u)) & 7u) + 32u)) {
                    if (!res_final) return suspend_input(tpos);
                    return fail(ST_TRUNCATED, 0, 0);
                }
                const int st = zlib_trailer();
                if (st) return st;
                phase = PH_DONE;
                return ST_OK;
            } else {
                return status;
            }
        }
    }

    // raw bytes in[p .. p + n) straight from HBM to the ring; the bit reader restarts behind them
    PZG_FN void stored_copy(uint64_t p, uint32_t n)
    {
        const uint32_t lane = lane_id();
        uint32_t done = 0;
        constexpr uint32_t PIECE = RING / 4u;
        while (done < n) {
            uint32_t piece = n - done < PIECE ? n - done : PIECE;
            if (op + piece - flushed > RING) flush_to(op & ~(uint64_t)15u);
#pragma nounroll
            for (uint32_t k0 = 0; k0 < piece; k0 += PZG_WAVE) {
                const uint32_t k = k0 + lane;
                const uint8_t v = in[p + done + (k < piece ? k : piece - 1u)];
                ring_store(k < piece, (((uint32_t)op + k)) & RMASK, v, lane);
            }
            op += piece;
            done += piece;
        }
        maybe_flush();
        in_byte0 = p + n;
        br.start(in, in_len, in_byte0);
    }

    // The wave's LDS image <-> its slot in HBM (SAVE: LDS -> HBM), the live part only: the ring's first min(produced, RING)
    // bytes (a decoder that has produced less than the ring holds has written ring[0 .. produced) and nothing else), then
    // everything behind the ring (tables, code lengths); the input prefetch buffer in front of the ring is dead between
    // calls.  A feed of a young stream moves ~4.5 KiB each way instead of 36.9 KiB.
    template <bool SAVE>
    PZG_FN void lds_image_copy(uint32_t *lds_image, uint64_t produced)
    {
        const uint32_t lane = lane_id();
        uint32_t *ldsw = (uint32_t *)(void *)&L;
        constexpr uint32_t NW = (uint32_t)(sizeof(WaveLds<RING_BITS>) / 4u);
        constexpr uint32_t RING0 = (uint32_t)(offsetof(WaveLds<RING_BITS>, ring) / 4u), RING1 = RING0 + RING / 4u;
        const uint32_t live1 = RING0 + (produced >= RING ? RING / 4u : ((uint32_t)produced + 3u) / 4u);
#pragma nounroll
        for (uint32_t k0 = RING0; k0 < live1; k0 += PZG_WAVE) {
            const uint32_t k = k0 + lane;
            if (k < live1) {
                if (SAVE) lds_image[k] = ldsw[k];
                else ldsw[k] = lds_image[k];
            }
        }
#pragma nounroll
        for (uint32_t k0 = RING1; k0 < NW; k0 += PZG_WAVE) {
            const uint32_t k = k0 + lane;
            if (k < NW) {
                if (SAVE) lds_image[k] = ldsw[k];
                else ldsw[k] = lds_image[k];
            }
        }
    }

    // One call of the resumable decoder on decoder state `rs` (+ its LDS image behind it in HBM).
    // hist: the decoder's 32 KiB history in HBM (small-ring instances only; unused, may be null, for the 32 KiB LDS ring)
    PZG_FN void run_resume(ResumeState *rs, uint32_t *lds_image, uint8_t *hist_, const uint8_t *in_, uint64_t in_len_, uint8_t *out_, uint64_t cap_,
                           uint32_t final_input, StreamResult *res, uint32_t *chunks_out)
    {
        hist = hist_;
        const uint32_t lane = lane_id();
        in = in_;
        in_len = in_len_;
        dict = nullptr;
        dict_len = 0;
        hist_extra = 0;
        gz_expect = 0;
        pend_m0 = pend_m1 = 0;
        pend_pos = 0;
        in_byte0 = 0;
        res_final = final_input;
        phase = rs->phase;
        bfinal_cur = rs->bfinal;
        stored_left = rs->stored_left;
        deferred = rs->deferred;
        qn = rs->qn;
        ow = rs->ow;
        chunks = rs->chunks;
        op = rs->op;
        flushed = op & ~(uint64_t)15u;  // flush_to() moves whole 16-byte groups: the last few bytes of a call wait in the ring for the next
        adler_a = rs->adler_a;
        adler_b = rs->adler_b;
        lit_e15 = rs->lit_e15;
        dist_e15 = rs->dist_e15;
        lit_n = rs->lit_n;
        dist_n = rs->dist_n;
        use_sub = rs->use_sub;
        lit_sub_used = rs->lit_sub_used;
        dist_sub_used = rs->dist_sub_used;
        s_rd = s_qn = s_total = s_cnt = s_lc = 0;  // (no span is ever under way between two calls)
        status = rs->status;
        detail0 = rs->detail0;
        detail1 = rs->detail1;
        const uint32_t bit_skip = rs->bit_skip;
        in_total_bits = rs->in_total * 8u;
        if (phase == PH_HEADER && op == 0u && adler_a == 0u) adler_a = 1;  // a fresh decoder (all zeros)
        const uint64_t flushed0 = flushed;
        out = out_ - flushed0;  // this call's room starts at produced-byte `flushed0`
        cap = flushed0 + cap_;
        set_far_base();
        PZG_LANES_BEGIN(j)
            PZG_LV(QT, j) = rs->QT[j];
        PZG_LANES_END
        // the LDS image (tables and the ring) as the previous call left it: the ring only as far as it has been written
        // (everything below 32 KiB of output sits at ring[0 .. op)), the prefetch buffer not at all
        {
            if (phase != PH_HEADER || op != 0u) {
                lds_image_copy<false>(lds_image, op);
            } else if (lane == 0u || PZG_WAVE == 1u) {
                L.fixed_ready = 0u;
            }
            wave_sync();
        }
#if PZG_DEVICE_PASS && PZG_DMA_PREFETCH
        br.pf = L.pf;
#endif
        br.start(in, in_len, 0);
        br.rp += bit_skip;  // (< 8: the partly consumed first byte)
        susp_pos = 0;
        int st = ST_OK;
        if (phase != PH_DONE) st = resume_decode();
        else st = status;
        if (st != ST_NEED_INPUT && st != ST_OUT_FULL) {
            if (st != ST_OK) status = st;
            phase = PH_DONE;
            susp_pos = stream_bit_pos();
        }
        complete_pending();
        flush_to(phase == PH_DONE ? op : op & ~(uint64_t)15u);  // what has been produced goes to this call's output (and into the checksum)
#if PZG_DEVICE_PASS && PZG_DMA_PREFETCH
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no fetch into this wave's LDS may outlive the call
#endif
        wave_sync();
        uint64_t used = susp_pos >> 3;
        if (used > in_len) used = in_len;
        {   // save
            lds_image_copy<true>(lds_image, op);
            PZG_LANES_BEGIN(j)
                rs->QT[j] = PZG_LV(QT, j);
            PZG_LANES_END
            if (lane == 0u || PZG_WAVE == 1u) {
                rs->phase = phase;
                rs->bfinal = bfinal_cur;
                rs->stored_left = stored_left;
                rs->deferred = deferred;
                rs->qn = qn;
                rs->bit_skip = phase == PH_DONE ? 0u : (uint32_t)(susp_pos & 7u);
                rs->ow = ow;
                rs->chunks = chunks;
                rs->op = op;
                rs->adler_a = adler_a;
                rs->adler_b = adler_b;
                rs->lit_e15 = lit_e15;
                rs->dist_e15 = dist_e15;
                rs->lit_n = lit_n;
                rs->dist_n = dist_n;
                rs->use_sub = use_sub;
                rs->lit_sub_used = lit_sub_used;
                rs->dist_sub_used = dist_sub_used;
                rs->status = status;
                rs->detail0 = detail0;
                rs->detail1 = detail1;
                rs->in_total += used;
            }
        }
        res->status = st == ST_NEED_INPUT || st == ST_OUT_FULL ? st : status;
        res->detail0 = detail0;
        res->detail1 = detail1;
        res->adler = (adler_b << 16) | adler_a;
        res->gz_crc = 0;
        res->out_len = flushed - flushed0;  // bytes delivered by THIS call
        res->in_used = used;
        *chunks_out = chunks;
    }
};

}  // namespace pzg
