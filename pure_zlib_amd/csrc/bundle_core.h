// bundle_core.h -- bundles (round 6): small zlib streams of the FIXED code, 64 to a wavefront, one lane per stream.
//
// One stream per wave (inflate_core.h) is the wrong grain for a stream of a few KiB: a 4 KiB level-1 stream (BASELINE config 3)
// is ONE span of 297-bit strips behind a 768-bit run-up -- 91 steps of guessing for 49 of decoding -- and its groups copy
// matches of four bytes with byte operations under lane masks that are mostly empty (measured: the LDS pipe is what that path,
// and a two-kernel split of it, run out of).  But a batch of small streams has what a strip lacks: 64 places where a token is
// KNOWN to start -- the streams' own first bits -- and blocks of the fixed code (Deflate.hs:79-82, 241-251) share their tables.
//
// So here lane k IS a sequential inflater for stream k (Deflate.hs:106-120 runInflate, one token a step):
//   * its input comes through a ring of eight-byte pairs in LDS that the wave refills for all lanes at once (no load in a step);
//   * its OutputWindow (OutputWindow.hs:29-114) is the last 512 bytes it produced, in LDS, dword-interleaved with the other lanes'
//     (dword d of lane k at win[(d & 127) * 64 + k]: every lane has a bank of its own, whatever its position) -- the window is
//     the lane's alone, so it is written by whole dwords (the partial last dword rides in a register and is rewritten every step);
//   * a token becomes a chunk of 1..8 bytes appended per step: a literal, 8 bytes of a near match (read as three aligned dwords
//     and funnel-shifted), or 8 bytes of a FAR match -- older than the window: read from the stream's own flushed output by a
//     load that is asked for in the step's first lines and used in its last;
//     a match longer than 8 bytes (or one that overlaps itself: the distance doubles as the pattern repeats) takes more steps;
//   * every PHASE steps the wave flushes the lanes' completed 16-byte groups to their outputs, folds them into the lanes'
//     Adler-32 (Adler32.hs:17-57), refills the input rings if a lane is down to half of its own, and finishes the lanes whose
//     stream has ended: tail bytes, trailer (Deflate.hs:52-63), results.
// A lane handles exactly what is plain: a valid zlib header without FDICT (Zlib.hs:55-68), blocks of the fixed code only,
// tokens inside the input, output inside the capacity, distances inside the output, a whole trailer.  ANYTHING else -- a stored
// or dynamic block, any error, a stream or capacity of a MiB or more -- leaves the stream to the ordinary kernel (BS_TODO), which
// decodes it from its first byte and reports what the reference reports.  Nothing about a result depends on which path produced
// it.  The same source compiles as a host program for the CPU model tests (wave.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "inflate_core.h"

namespace pzg {

struct BundleLds {
    static constexpr uint32_t BQ_PAIRS = 8u;   // input ring, pairs per lane
    static constexpr uint32_t WIN_DW = 128u;   // output window, dwords per lane (512 bytes)
    uint32_t lit[512];                         // HuffmanTree of the fixed literal/length code: one 2^9-entry table
    uint32_t dist[32];                         // ... of the fixed distance code
    uint64_t bq[BQ_PAIRS * 64u];               // bq[slot * 64 + lane]
    uint32_t win[WIN_DW * 64u];                // win[(dword & 127) * 64 + lane]
};

struct Bundle {
    enum : uint32_t { BS_TODO = 0, BS_CLEAN = 1, BS_FIN = 2, BS_RUN = 3 };
    static constexpr uint32_t MAX_BYTES = 1u << 20;       // streams and capacities a lane's 32-bit positions hold with room to spare
    static constexpr uint32_t BQ_PAIRS = BundleLds::BQ_PAIRS, WIN_DW = BundleLds::WIN_DW, WIN_BYTES = 4u * WIN_DW;
    static constexpr uint32_t NEAR_MAX = WIN_BYTES - 16u; // a distance up to this is read from the window (the append may clobber the 12 oldest bytes)
    static constexpr uint32_t PHASE = 4u;                 // steps between two looks at the rings and the flushes
#ifndef PZG_BUNDLE_BQ_LOW
#define PZG_BUNDLE_BQ_LOW 5
#endif
    // A lane with fewer pairs than this in its ring at a phase's start has the rings refilled.  A step consumes 49 bits at the most and
    // moves a lane on once at the most: four pairs a phase -- with five, the pair a lane looks at next (NXT) has always been written.
    // (Four, and NXT read again after a refill: 5 % slower, measured -- an LDS read right behind the refill's writes.)
    static constexpr uint32_t BQ_LOW = PZG_BUNDLE_BQ_LOW;
    static_assert(BQ_LOW >= PHASE + 1u && BQ_LOW <= BQ_PAIRS, "the ring never runs dry inside a phase");
    static constexpr uint32_t TK_MATCH = ENT_MATCH;

    struct In {                            // lane k's stream (ON = 0: none)
        LaneVec<const uint8_t *> IN;
        LaneVec<uint8_t *> OUT;
        LaneVec<uint32_t> LEN, CAP, ON;    // compressed bytes, output capacity (both below MAX_BYTES for a lane that is ON)
    };
    struct Out {
        LaneVec<uint32_t> STATE;           // BS_CLEAN: decoded, the result words below are the stream's; BS_TODO: the ordinary kernel's
        LaneVec<uint32_t> STATUS, D0, D1, ADLER, OLEN, USED;
    };

    // ---- the fixed code's tables (Deflate.hs:241-251), arithmetically: entry of the 9 / 5 stream bits `idx` -------------------
    PZG_FN static uint32_t fixed_lit_entry(uint32_t idx)
    {
        const uint32_t c = bitrev32(idx) >> 23;  // the nine bits as a code, first bit first
        if ((c >> 2) < 24u) return litlen_entry(256u + (c >> 2), 7u);                 // 0000000 .. 0010111
        if ((c >> 1) < 192u) return litlen_entry((c >> 1) - 48u, 8u);                 // 00110000 .. 10111111
        if ((c >> 1) < 200u) return litlen_entry(280u + ((c >> 1) - 192u), 8u);       // 11000000 .. 11000111
        return litlen_entry(144u + (c - 400u), 9u);                                   // 110010000 .. 111111111
    }
    PZG_FN static uint32_t fixed_dist_entry(uint32_t idx) { return dist_entry(bitrev32(idx) >> 27, 5u); }
    PZG_FN static void build_tables(BundleLds &L)
    {
        PZG_LANES_BEGIN(k)
            for (uint32_t j = 0; j < 8u; ++j) L.lit[64u * j + k] = fixed_lit_entry(64u * j + k);
            if (k < 32u) L.dist[k] = fixed_dist_entry(k);
        PZG_LANES_END
        wave_sync();
    }

    PZG_FN static uint64_t pair_at(const uint32_t *sp, uint32_t i)
    {
        const uint32_t *q = (const uint32_t *)(const void *)((const uint8_t *)(const void *)sp + (i << 2));
        return (uint64_t)q[0] | ((uint64_t)q[1] << 32);
    }
    // up to BQ_PAIRS more pairs of a lane's input into its ring (pairs past the stream's last dwords repeat that pair: no token reaches there)
    PZG_FN static void refill(const uint32_t *sp, uint32_t maxdw, uint64_t *bq, uint32_t k, uint32_t &nx, uint32_t rdi, uint32_t &wri)
    {
        const uint32_t n = BQ_PAIRS - (wri - rdi);
        uint64_t v[BQ_PAIRS];
#pragma unroll
        for (uint32_t j = 0; j < BQ_PAIRS; ++j) {
            const uint32_t i = nx + 2u * j;
            v[j] = pair_at(sp, (j < n && i < maxdw) ? i : maxdw);
        }
#pragma unroll
        for (uint32_t j = 0; j < BQ_PAIRS; ++j)
            if (j < n) bq[((wri + j) & (BQ_PAIRS - 1u)) * 64u + k] = v[j];
        wri += n;
        nx += 2u * n;
    }
    // x % 65521 for x < 2^32 (Adler32.hs:22-27)
    PZG_FN static uint32_t mod_adler(uint32_t x)
    {
        const uint32_t q = (uint32_t)(((uint64_t)x * 0x80078071ull) >> 47);
        return x - q * ADLER_MOD;
    }
    PZG_FN static uint32_t win_addr(uint32_t dw, uint32_t k) { return ((dw & (WIN_DW - 1u)) << 6) + k; }

    struct State {
        LaneVec<const uint32_t *> SP;
        LaneVec<uint64_t> W0, W1, NXT, FARV[2];   // FARV: the far loads' landing registers (they alternate with the steps)
        LaneVec<uint32_t> MAXDW, R, NX, RD, WR;       // the reader
        LaneVec<uint32_t> P, END, ST, BF;             // bit position, end of the input, BS_*, the block is the last one
        LaneVec<uint32_t> OB, FL, ACC;                // bytes produced, bytes flushed (a multiple of 16), the partial last dword
        LaneVec<uint32_t> ML, MD;                     // the match under way: bytes still to copy, its (effective) distance
        LaneVec<uint32_t> FN[2];                      // bytes the far load of this slot brings (0: none on its way)
        LaneVec<uint32_t> AA, AB;                     // Adler-32 of the flushed bytes
        LaneVec<uint32_t> ENDP;                       // (BS_FIN) where the trailer starts, in bits
    };

    // Decodes the bundle.  `common`: sixteen readable bytes, 4-byte aligned (what idle lanes load).
    PZG_FN static void run(BundleLds &L, const In &bi, const uint32_t *common, Out &bo)
    {
        build_tables(L);
        State s;
        PZG_LANES_BEGIN(k)
            const uint32_t len = PZG_LV(bi.LEN, k), cap = PZG_LV(bi.CAP, k);
            const bool on = PZG_LV(bi.ON, k) != 0u && len >= 8u && len < MAX_BYTES && cap < MAX_BYTES;
            const uint8_t *p = on ? PZG_LV(bi.IN, k) : (const uint8_t *)(const void *)common;
            const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
            const uint32_t *sp = (const uint32_t *)(const void *)(p - mis);
            const uint32_t ndw = on ? (mis + len + 3u) >> 2 : 2u;
            const uint32_t maxdw = (ndw - 2u) & ~1u;
            PZG_LV(s.SP, k) = sp;
            PZG_LV(s.MAXDW, k) = maxdw;
            PZG_LV(s.END, k) = on ? 8u * (mis + len) : 0u;
            PZG_LV(s.W0, k) = pair_at(sp, 0u);
            PZG_LV(s.W1, k) = pair_at(sp, 2u < maxdw ? 2u : maxdw);
            PZG_LV(s.NX, k) = 4u;
            PZG_LV(s.RD, k) = 0u;
            PZG_LV(s.WR, k) = 0u;
            refill(sp, maxdw, L.bq, k, PZG_LV(s.NX, k), PZG_LV(s.RD, k), PZG_LV(s.WR, k));
            PZG_LV(s.NXT, k) = L.bq[k];
            // Zlib.hs:55-68: CMF, FLG -- FCHECK, CM = 8, CINFO <= 7, no FDICT -- and the first block's three bits (Deflate.hs:67-68)
            const uint32_t hw = (uint32_t)(PZG_LV(s.W0, k) >> (8u * mis));  // (mis <= 3: the 19 bits lie in W0)
            const uint32_t cmf = hw & 0xffu, flg = (hw >> 8) & 0xffu, bh = (hw >> 16) & 7u;
            const bool plain = ((cmf << 8) | flg) % 31u == 0u && (cmf & 15u) == 8u && (cmf >> 4) <= 7u && (flg & 0x20u) == 0u && (bh >> 1) == 1u;
            PZG_LV(s.ST, k) = (on && plain) ? (uint32_t)BS_RUN : (uint32_t)BS_TODO;
            PZG_LV(s.BF, k) = bh & 1u;
            PZG_LV(s.P, k) = 8u * mis + 19u;
            PZG_LV(s.R, k) = 8u * mis + 19u;
            PZG_LV(s.OB, k) = PZG_LV(s.FL, k) = PZG_LV(s.ACC, k) = 0u;
            PZG_LV(s.ML, k) = PZG_LV(s.MD, k) = 0u;
            PZG_LV(s.FARV[0], k) = PZG_LV(s.FARV[1], k) = 0ull;
            PZG_LV(s.FN[0], k) = PZG_LV(s.FN[1], k) = 0u;
            PZG_LV(s.AA, k) = 1u;
            PZG_LV(s.AB, k) = 0u;
            PZG_LV(s.ENDP, k) = 0u;
            PZG_LV(bo.STATE, k) = BS_TODO;
            PZG_LV(bo.STATUS, k) = PZG_LV(bo.D0, k) = PZG_LV(bo.D1, k) = PZG_LV(bo.ADLER, k) = PZG_LV(bo.OLEN, k) = PZG_LV(bo.USED, k) = 0u;
        PZG_LANES_END
        for (;;) {
            // ---- the phase: everything that touches memory but the far loads ---------------------------------------------------
            {
                LaneVec<bool> LIVE, LOW;
                PZG_LANES_BEGIN(k)
                    const uint32_t st = PZG_LV(s.ST, k);
                    PZG_LV(LIVE, k) = (st == (uint32_t)BS_RUN) | (st == (uint32_t)BS_FIN);
                    PZG_LV(LOW, k) = (st == (uint32_t)BS_RUN) & (PZG_LV(s.WR, k) - PZG_LV(s.RD, k) < BQ_LOW);
                PZG_LANES_END
                if (lanes_ballot(LIVE) == 0ull) break;
#if PZG_DEVICE_PASS
                // every store of the phases before this one has landed from here on: a far load of the steps that follow reads bytes
                // that were flushed two phases ago or earlier (a far source ends 488 bytes or more behind the lane's position, a lane
                // produces at most 8 * PHASE bytes between two flushes)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                if (lanes_ballot(LOW) != 0ull) {
                    PZG_LANES_BEGIN(k)
                        refill(PZG_LV(s.SP, k), PZG_LV(s.MAXDW, k), L.bq, k, PZG_LV(s.NX, k), PZG_LV(s.RD, k), PZG_LV(s.WR, k));
                    PZG_LANES_END
                }
                flush_groups(L, bi, s);
                finish_lanes(L, bi, s, bo);
            }
#pragma unroll
            for (uint32_t u0 = 0; u0 < PHASE; ++u0) step(L, bi, common, s, u0 & 1u);  // (unrolled: the landing registers alternate)
        }
    }

    // the lanes' completed 16-byte groups: window -> output, folded into the lanes' Adler-32
    PZG_FN static void flush_groups(BundleLds &L, const In &bi, State &s)
    {
        for (;;) {
            LaneVec<bool> DUE;
            PZG_LANES_BEGIN(k)
                const uint32_t st = PZG_LV(s.ST, k);
                PZG_LV(DUE, k) = ((st == (uint32_t)BS_RUN) | (st == (uint32_t)BS_FIN)) & (PZG_LV(s.FL, k) + 16u <= PZG_LV(s.OB, k));
            PZG_LANES_END
            if (lanes_ballot(DUE) == 0ull) break;
            PZG_LANES_BEGIN(k)
                if (PZG_LV(DUE, k)) {
                    const uint32_t fl = PZG_LV(s.FL, k), a0 = win_addr(fl >> 2, k);  // (a group's four dwords do not wrap: 16 divides 512)
                    const uint32_t x0 = L.win[a0], x1 = L.win[a0 + 64u], x2 = L.win[a0 + 128u], x3 = L.win[a0 + 192u];
                    uint8_t *o = PZG_LV(bi.OUT, k) + fl;
#if PZG_DEVICE_PASS
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    typedef u32x4 __attribute__((aligned(1))) u32x4_u;
                    const u32x4 v = {x0, x1, x2, x3};
                    *(u32x4_u *)(void *)o = v;
#else
                    const uint32_t xs[4] = {x0, x1, x2, x3};
                    __builtin_memcpy(o, xs, 16);
#endif
                    // Adler32.hs:29-34 advanceNoMod over 16 bytes at once: s = sum d_i, t = sum (16-i) d_i
                    const uint32_t sm = sum4(x0, sum4(x1, sum4(x2, sum4(x3, 0u))));
                    const uint32_t t = dot4(x0, 0x0D0E0F10u, dot4(x1, 0x090A0B0Cu, dot4(x2, 0x05060708u, dot4(x3, 0x01020304u, 0u))));
                    PZG_LV(s.AB, k) = mod_adler(PZG_LV(s.AB, k) + 16u * PZG_LV(s.AA, k) + t);
                    PZG_LV(s.AA, k) = mod_adler(PZG_LV(s.AA, k) + sm);
                    PZG_LV(s.FL, k) = fl + 16u;
                }
            PZG_LANES_END
        }
    }
    // the lanes whose stream has ended (BS_FIN; everything but the last few bytes is flushed): tail, trailer, results
    PZG_FN static void finish_lanes(BundleLds &L, const In &bi, State &s, Out &bo)
    {
        LaneVec<bool> FIN;
        PZG_LANES_BEGIN(k)
            PZG_LV(FIN, k) = PZG_LV(s.ST, k) == (uint32_t)BS_FIN;
        PZG_LANES_END
        if (lanes_ballot(FIN) == 0ull) return;
        PZG_LANES_BEGIN(k)
            if (PZG_LV(FIN, k)) {
                const uint32_t fl = PZG_LV(s.FL, k), ob = PZG_LV(s.OB, k);
                uint32_t a = PZG_LV(s.AA, k), b = PZG_LV(s.AB, k);
                for (uint32_t t = 0; t < 15u; ++t) {  // (Adler32.hs:22-27, byte by byte: fewer than 16 are left)
                    if (fl + t < ob) {
                        const uint32_t pos = fl + t;
                        const uint8_t d = (uint8_t)(L.win[win_addr(pos >> 2, k)] >> (8u * (pos & 3u)));
                        PZG_LV(bi.OUT, k)[pos] = d;
                        a += d;
                        b += a;
                    }
                }
                a = mod_adler(a);
                b = mod_adler(b);
                const uint32_t ours = (b << 16) | a;
                const uint32_t endp = PZG_LV(s.ENDP, k);
                const uint8_t *tp = (const uint8_t *)(const void *)PZG_LV(s.SP, k) + (endp >> 3);  // big-endian (Monad.hs:257-263)
                const uint32_t theirs = ((uint32_t)tp[0] << 24) | ((uint32_t)tp[1] << 16) | ((uint32_t)tp[2] << 8) | (uint32_t)tp[3];
                const uint32_t mis8 = 8u * (uint32_t)((uintptr_t)PZG_LV(bi.IN, k) & 3u);
                const bool good = theirs == ours;
                PZG_LV(bo.STATE, k) = BS_CLEAN;
                PZG_LV(bo.STATUS, k) = good ? (uint32_t)ST_OK : (uint32_t)ST_CHECKSUM;  // Deflate.hs:52-63 checkChecksum
                PZG_LV(bo.D0, k) = good ? 0u : theirs;
                PZG_LV(bo.D1, k) = good ? 0u : ours;
                PZG_LV(bo.ADLER, k) = ours;
                PZG_LV(bo.OLEN, k) = ob;
                PZG_LV(bo.USED, k) = ((endp - mis8) >> 3) + 4u;
                PZG_LV(s.ST, k) = BS_CLEAN;
            }
        PZG_LANES_END
    }

    // One step.  A lane with a match under way copies up to eight bytes of it (or asks for them, or takes what a far load
    // brought); a lane without one decodes up to three literals -- three tokens in four of config 3's streams are literals; a
    // literal of the fixed code is 8 or 9 bits, so the tokens behind the first are looked up at all the places they can start, in
    // the same trip to the LDS as the first -- and the match behind them, whose first bytes go out in the same step as the
    // literals: a chunk is 1..8 bytes (measured on config 3's streams: 1,183 steps for the longest of 64 where one token a step
    // takes 2,504 and two literals or one match 1,609).  The distance code of the fixed code is five bits in reverse:
    // arithmetic, no table -- a step is two trips to the LDS, the tables and then the window.  A match older than the window (FAR: nearly
    // half of config 3's) is read from the stream's own flushed output, eight bytes a step: a step asks for the bytes that the NEXT step
    // appends -- into one of two landing registers that alternate with the steps (u: this step's; the wave is alone on its SIMD, so a
    // load that is used in the step that asks for it is waited for in full: measured).
    PZG_FN static bool is_lit(uint32_t e) { return (e & (ENT_MATCH | ENT_STOP)) == 0u; }
    PZG_FN static void step(BundleLds &L, const In &bi, const uint32_t *common, State &s, uint32_t u)
    {
        LaneVec<uint32_t> TKS, WLO;
        LaneVec<bool> STOPF;
        PZG_MARK("bs.begin");
        PZG_LANES_BEGIN(k)
            // ---- the reader: 64 bits on if the position says so (the next pair waits in NXT, the one behind it is asked for)
            {
                const bool sh = PZG_LV(s.R, k) >= 64u;
                PZG_LV(s.W0, k) = sh ? PZG_LV(s.W1, k) : PZG_LV(s.W0, k);
                PZG_LV(s.W1, k) = sh ? PZG_LV(s.NXT, k) : PZG_LV(s.W1, k);
                PZG_LV(s.RD, k) += sh ? 1u : 0u;
                PZG_LV(s.R, k) &= 63u;
                PZG_LV(s.NXT, k) = L.bq[(PZG_LV(s.RD, k) & (BQ_PAIRS - 1u)) * 64u + k];
            }
            const uint32_t ob = PZG_LV(s.OB, k);
            const uint32_t ml0 = PZG_LV(s.ML, k), md0 = PZG_LV(s.MD, k);
            const bool run = PZG_LV(s.ST, k) == (uint32_t)BS_RUN;
            const uint32_t fnc = PZG_LV(s.FN[u ^ 1u], k);  // bytes of a far match that the step before this one asked for: they go in now
            // ---- every LDS read of the step: the tokens' entries, the window at the armed match's source
            uint32_t wlo, whi;
            {
                const uint64_t w0 = PZG_LV(s.W0, k), w1 = PZG_LV(s.W1, k);
                const uint32_t b0 = (uint32_t)w0, b1 = (uint32_t)(w0 >> 32), b2 = (uint32_t)w1, b3 = (uint32_t)(w1 >> 32), r = PZG_LV(s.R, k);
                const bool up = r >= 32u;
                const uint32_t lo = up ? b1 : b0, mid = up ? b2 : b1, hi = up ? b3 : b2;
                wlo = funnel(mid, lo, r);
                whi = funnel(hi, mid, r);
            }
            const uint32_t e1 = L.lit[wlo & 511u];
            const uint32_t e8 = L.lit[(wlo >> 8) & 511u], e9 = L.lit[(wlo >> 9) & 511u];
            const uint32_t e16 = L.lit[(wlo >> 16) & 511u], e17 = L.lit[(wlo >> 17) & 511u], e18 = L.lit[(wlo >> 18) & 511u];
            const bool dec = run & (ml0 == 0u);
            // ---- the tokens (Monad.hs:295-302 nextCode, Deflate.hs:106-120): literals 1..3, then the token to arm
            const bool l1 = is_lit(e1);
            const uint32_t n1 = e1 & 31u, e2 = (n1 & 1u) ? e9 : e8;
            const bool l2 = l1 & is_lit(e2);
            const uint32_t n12 = n1 + (e2 & 31u), e3 = n12 == 16u ? e16 : n12 == 17u ? e17 : e18;
            const bool l3 = l2 & is_lit(e3);
            const uint32_t n123 = n12 + (e3 & 31u);
            const uint32_t pos = PZG_LV(s.P, k), end = PZG_LV(s.END, k), cap = PZG_LV(bi.CAP, k);
            // (a literal is taken if it lies inside the input and the capacity; one that does not is the next step's first token)
            const bool t1 = dec & l1 & (pos + n1 <= end) & (ob + 1u <= cap);
            const bool t2 = t1 & l2 & (pos + n12 <= end) & (ob + 2u <= cap);
            const bool t3 = t2 & l3 & (pos + n123 <= end) & (ob + 3u <= cap);
            const uint32_t nl = t3 ? 3u : t2 ? 2u : t1 ? 1u : 0u;
            // the token behind the literals that were taken (behind all three: none)
            const uint32_t ea = !l1 ? e1 : !l2 ? e2 : e3, pa = !l1 ? 0u : !l2 ? n1 : n12;
            const bool first = !l1;                                  // the token to arm is the step's first
            const bool reach = first | (!l2 ? t1 : !l3 ? t2 : false);  // every literal in front of it was taken
            const bool am = (int32_t)ea < 0;                         // a length: a match
            const uint32_t wa = funnel(whi, wlo, pa);                // 32 bits from its first (pa <= 18; length + distance are 31 bits at the most)
            const uint32_t na = (ea >> 8) & 31u, ta = ea & 31u;
            const uint32_t mlen = ((ea >> 16) & 511u) + ubfe(wa, na, ta - na);
            const uint32_t wd = wa >> ta;
            const uint32_t dsym = bitrev32(wd) >> 27;                // the fixed distance code: five bits, first bit first (Deflate.hs:249-251)
            const uint32_t de = dsym >= 4u ? (dsym >> 1) - 1u : 0u;
            const uint32_t dbase = dsym >= 4u ? 1u + ((2u + (dsym & 1u)) << de) : 1u + dsym;  // Deflate.hs:203-237 distanceArray
            const uint32_t mdist = dbase + ubfe(wd, 5u, de);
            const uint32_t tba = ta + 5u + de;
            const uint32_t bits_l = t3 ? n123 : t2 ? n12 : t1 ? n1 : 0u;
            // what sends the stream to the ordinary kernel, when it is the step's FIRST token: a token that reaches past the input's
            // end, output past the capacity, a distance in front of the output, a symbol that is none (behind literals it is left
            // where it is and becomes a step's first token)
            const bool mfit = (pos + pa + tba <= end) & (ob + nl + mlen <= cap) & (mdist <= ob + nl) & (dsym < 30u);
            const bool arm = dec & reach & am & mfit;
            const bool stop1 = dec & first & ((e1 & ENT_STOP) != 0u);
            const bool bad = dec & ((first & am & !mfit) | (l1 & !t1));
            PZG_LV(s.ST, k) = bad ? (uint32_t)BS_TODO : PZG_LV(s.ST, k);
            PZG_LV(STOPF, k) = stop1;
            PZG_LV(TKS, k) = e1;
            PZG_LV(WLO, k) = wlo;
            const uint32_t adv = bits_l + (arm ? tba : 0u);
            PZG_LV(s.P, k) = pos + adv;
            PZG_LV(s.R, k) += adv;
            // ---- the near match under way (one that was just decoded: behind the step's literals): three aligned dwords of the
            // lane's window, funnel-shifted (OutputWindow.hs:82-101: a piece of at most `distance` bytes, and nothing of what this
            // step appends)
            const uint32_t ml = arm ? mlen : ml0, md = arm ? mdist : md0;
            const bool nearm = run & (ml != 0u) & (md <= NEAR_MAX);
            const uint32_t room = 8u - nl, reach_b = md > nl ? md - nl : 0u;  // (nl = 0 unless the match is the step's own)
            uint32_t nn = ml < room ? ml : room;
            nn = nn < reach_b ? nn : reach_b;
            nn = nearm ? nn : 0u;
            const uint32_t src = ob + nl - md, sdw = src >> 2, so = (src & 3u) << 3;
            const uint32_t s0 = L.win[win_addr(sdw, k)], s1 = L.win[win_addr(sdw + 1u, k)], s2 = L.win[win_addr(sdw + 2u, k)];
            const uint32_t nlo = funnel(s1, s0, so), nhi = funnel(s2, s1, so);
            // ---- a far match under way (or just decoded): the bytes behind what this step appends are asked for
            {
                const uint32_t rem = ml - fnc;  // (fnc != 0 only while a far match is under way: ml = ml0 >= fnc)
                const bool farq = run & (md > NEAR_MAX) & (rem != 0u);
                const uint8_t *fp = farq ? PZG_LV(bi.OUT, k) + (ob + nl + fnc - md) : (const uint8_t *)(const void *)common;
                const uint32_t fnew = farq ? (rem < 8u ? rem : 8u) : 0u;
#if PZG_DEVICE_PASS
                typedef uint64_t __attribute__((aligned(1))) u64_u;
                PZG_LV(s.FARV[u], k) = *(const u64_u *)(const void *)fp;
#else
                uint64_t v = 0;
                __builtin_memcpy(&v, fp, farq ? fnew : 8u);  // (the model reads no byte it does not use)
                PZG_LV(s.FARV[u], k) = v;
#endif
                PZG_LV(s.FN[u], k) = fnew;
            }
            // ---- the chunk: n bytes of data -- what the far load of the step before brought, or the step's literals with the match's
            // bytes behind them
            const bool land = fnc != 0u;
            const uint64_t fv = PZG_LV(s.FARV[u ^ 1u], k);
            const uint32_t n = land ? fnc : nl + nn;
            // (what lies behind the chunk's n-th byte in `data` means nothing and is not masked away: it goes to window positions that
            // are yet to be produced, and to ACC's bytes past the position, which the next append cuts off)
            const uint32_t lits = ubfe(((e1 >> 8) & 0xffu) | (e2 & 0xff00u) | ((e3 & 0xff00u) << 8), 0u, nl << 3);
            const uint64_t nd = (((uint64_t)nhi << 32) | nlo) << (nl << 3);
            const uint32_t dlo = land ? (uint32_t)fv : lits | (uint32_t)nd;
            const uint32_t dhi = land ? (uint32_t)(fv >> 32) : (uint32_t)(nd >> 32);
            // (an overlapping match: once a whole period is out, what was appended repeats the pattern and the distance may double --
            // Monad.hs:324-333 copies `distance` bytes at a time, which comes to the same bytes)
            PZG_LV(s.ML, k) = ml - (land ? fnc : nn);
            PZG_LV(s.MD, k) = (nearm & (md < 8u) & (nn == md)) ? md << 1 : md;
            // ---- the append: the partial last dword rides in ACC; three dwords are written whatever n is (the two behind the
            // first hold the window's oldest bytes, which no near match reaches: NEAR_MAX)
            {
                const uint32_t sh = (ob & 3u) << 3;
                const uint64_t x = (((uint64_t)dhi << 32) | dlo) << sh;
                const uint32_t d0 = ubfe(PZG_LV(s.ACC, k), 0u, sh) | (uint32_t)x, d1 = (uint32_t)(x >> 32), d2 = (uint32_t)(((uint64_t)dhi << sh) >> 32);
                const uint32_t odw = ob >> 2, tbts = (ob & 3u) + n, c = tbts >> 2;
                L.win[win_addr(odw, k)] = d0;
                L.win[win_addr(odw + 1u, k)] = d1;
                L.win[win_addr(odw + 2u, k)] = d2;
                PZG_LV(s.ACC, k) = c == 0u ? d0 : c == 1u ? d1 : d2;
                PZG_LV(s.OB, k) = ob + n;
            }
        PZG_LANES_END
        PZG_MARK("bs.end");
        if (__builtin_expect(lanes_ballot(STOPF) != 0ull, 0)) {
            // the end of a block (Deflate.hs:45-50): the stream's end if the block was final, else the next block's three bits -- and
            // anything that is not the end-of-block code or not followed by a block of the fixed code is not a bundle's business
            PZG_LANES_BEGIN(k)
                if (PZG_LV(STOPF, k)) {
                    const uint32_t tk = PZG_LV(TKS, k), n = tk & 31u, pe = PZG_LV(s.P, k) + n;
                    const bool eob = (int32_t)tk >= 0 && ((tk >> 8) & 15u) == (uint32_t)K_EOB && pe <= PZG_LV(s.END, k);
                    uint32_t st = BS_TODO;
                    if (eob && PZG_LV(s.BF, k) != 0u) {
                        const uint32_t al = (pe + 7u) & ~7u;  // advanceToByte, then the four bytes of the trailer (Deflate.hs:52-63)
                        if (al + 32u <= PZG_LV(s.END, k)) {
                            st = BS_FIN;
                            PZG_LV(s.ENDP, k) = al;
                        }
                    } else if (eob) {
                        const uint32_t bh = (PZG_LV(WLO, k) >> n) & 7u;
                        if (pe + 3u <= PZG_LV(s.END, k) && (bh >> 1) == 1u) {
                            st = BS_RUN;
                            PZG_LV(s.BF, k) = bh & 1u;
                            PZG_LV(s.P, k) = pe + 3u;
                            PZG_LV(s.R, k) += n + 3u;
                        }
                    }
                    PZG_LV(s.ST, k) = st;
                }
            PZG_LANES_END
        }
    }
};

}  // namespace pzg
