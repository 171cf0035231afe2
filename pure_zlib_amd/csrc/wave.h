// wave.h -- the few wavefront primitives the inflate/adler kernels are written against.
//
// Device build (hipcc, gfx950): a wave is 64 lanes; wave-uniform values are pinned into
// SGPRs with v_readfirstlane so the decode state machine runs on the scalar unit.
//
// Host build (plain g++, tests/model only): PZG_WAVE == 1.  The same source then executes
// as a single-lane program, which lets the CPU test-suite fuzz the kernels' control logic,
// table construction and error ordering without a GPU.  The host build is test
// infrastructure: libpzg.so never contains or calls it (there is no CPU fallback).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PZG_FN __host__ __device__ __forceinline__
#else
#define PZG_FN inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PZG_DEVICE_PASS 1
#define PZG_WAVE 64u
#else
#define PZG_DEVICE_PASS 0
#define PZG_WAVE 1u
#endif

namespace pzg {

PZG_FN uint32_t lane_id()
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#else
    return 0u;
#endif
}

// wave-uniform value -> SGPR
PZG_FN uint32_t uni(uint32_t x)
{
#if PZG_DEVICE_PASS
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
#else
    return x;
#endif
}

PZG_FN uint64_t uni64(uint64_t x)
{
#if PZG_DEVICE_PASS
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
#else
    return x;
#endif
}

// value held by lane `l` (l wave-uniform)
PZG_FN uint32_t read_lane(uint32_t v, uint32_t l)
{
#if PZG_DEVICE_PASS
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
#else
    (void)l;
    return v;
#endif
}

PZG_FN uint64_t ballot(bool p)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_ballot_w64(p);
#else
    return p ? 1ull : 0ull;
#endif
}

// number of set bits of m below this lane
PZG_FN uint32_t mbcnt(uint64_t m)
{
#if PZG_DEVICE_PASS
    // (a mask known to be empty at compile time -- queue_append() for one half only -- counts nothing: the compiler does not
    // fold the two instructions itself, it hoists their result to the kernel's prologue and keeps it in a register, or
    // spills it: the one scratch slot of round 2's build)
    if (__builtin_constant_p(m) && m == 0ull) return 0u;
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    (void)m;
    return 0u;
#endif
}

PZG_FN uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }

// Orders this wave's LDS traffic between cooperative phases.  The workgroup is exactly one
// wave, so this is a compiler + lgkmcnt fence, not a multi-wave rendezvous.
PZG_FN void wave_sync()
{
#if PZG_DEVICE_PASS
    __syncthreads();
#endif
}

#if PZG_DEVICE_PASS
// DPP controls (gfx9/CDNA): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_zero(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
#endif

// sum over the wave, result uniform
PZG_FN uint32_t wave_sum(uint32_t x)
{
#if PZG_DEVICE_PASS
    // the inclusive scan's six DPP additions, the total read from lane 63 (a crossbar butterfly costs six instructions a step)
    x += dpp_zero<0x111, 0xf>(x);
    x += dpp_zero<0x112, 0xf>(x);
    x += dpp_zero<0x114, 0xf>(x);
    x += dpp_zero<0x118, 0xf>(x);
    x += dpp_zero<0x142, 0xa>(x);
    x += dpp_zero<0x143, 0xc>(x);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
#else
    return x;
#endif
}

// inclusive prefix sum over the lanes of a per-lane value (the one-lane host model: the value itself)
PZG_FN uint32_t wave_iscan_add(uint32_t x)
{
#if PZG_DEVICE_PASS
    x += dpp_zero<0x111, 0xf>(x);
    x += dpp_zero<0x112, 0xf>(x);
    x += dpp_zero<0x114, 0xf>(x);
    x += dpp_zero<0x118, 0xf>(x);
    x += dpp_zero<0x142, 0xa>(x);
    x += dpp_zero<0x143, 0xc>(x);
#endif
    return x;
}

PZG_FN uint32_t wave_max(uint32_t x)
{
#if PZG_DEVICE_PASS
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t y = (uint32_t)__shfl_xor((int)x, o);
        x = x > y ? x : y;
    }
    return uni(x);
#else
    return x;
#endif
}

// sum of the four bytes of x, plus c
PZG_FN uint32_t sum4(uint32_t x, uint32_t c)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_udot4(x, 0x01010101u, c, false);
#else
    return c + (x & 0xff) + ((x >> 8) & 0xff) + ((x >> 16) & 0xff) + (x >> 24);
#endif
}

// dot product of the four bytes of x with the four bytes of w, plus c
PZG_FN uint32_t dot4(uint32_t x, uint32_t w, uint32_t c)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_udot4(x, w, c, false);
#else
    return c + (x & 0xff) * (w & 0xff) + ((x >> 8) & 0xff) * ((w >> 8) & 0xff) +
           ((x >> 16) & 0xff) * ((w >> 16) & 0xff) + (x >> 24) * (w >> 24);
#endif
}

// ---- per-lane values -------------------------------------------------------------------------
// Code between PZG_LANES_BEGIN(k) and PZG_LANES_END runs once per lane k of a 64-lane wave: on the
// device it is the ordinary SIMT body (k = lane id, a LaneVec is one VGPR); in the one-thread host
// model it is an explicit loop over 64 lanes and a LaneVec is a 64-entry array.  Only code that
// does not communicate between lanes inside the block may be written this way.
#if PZG_DEVICE_PASS
template <class T>
struct LaneVec {
    T v;
};
#define PZG_LANES_BEGIN(k) { [[maybe_unused]] const uint32_t k = ::pzg::lane_id();
#define PZG_LANES_END }
#define PZG_LV(x, k) ((x).v)
#else
template <class T>
struct LaneVec {
    T v[64];
};
#define PZG_LANES_BEGIN(k) for (uint32_t k = 0; k < 64u; ++k) {
#define PZG_LANES_END }
#define PZG_LV(x, k) ((x).v[k])
#endif

// value of lane `l` (wave-uniform l) of a LaneVec
PZG_FN uint32_t lane_get(const LaneVec<uint32_t> &x, uint32_t l)
{
#if PZG_DEVICE_PASS
    return (uint32_t)__builtin_amdgcn_readlane((int)x.v, (int)l);
#else
    return x.v[l & 63u];
#endif
}

PZG_FN uint64_t lane_get64(const LaneVec<uint64_t> &x, uint32_t l)
{
#if PZG_DEVICE_PASS
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x.v, (int)l) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x.v >> 32), (int)l) << 32);
#else
    return x.v[l & 63u];
#endif
}

// bit k of a wave-uniform mask, as lane k's predicate: on the device the mask itself is the select/branch condition
// (inverse ballot), no shift and no compare
PZG_FN bool lane_bit(uint64_t m, uint32_t k)
{
#if PZG_DEVICE_PASS
    (void)k;
    return __builtin_amdgcn_inverse_ballot_w64(m);
#else
    return ((m >> k) & 1ull) != 0;
#endif
}

// bitwise m ? a : b (ONE three-input bit operation, v_bitop3_b32 with truth table 0xca)
PZG_FN uint32_t bit_select(uint32_t m, uint32_t a, uint32_t b)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_bitop3_b32(m, a, b, 0xca);
#else
    return (a & m) | (b & ~m);
#endif
}

// bit k of the wave-uniform mask m ? a : b, as ONE v_cndmask with the mask as its scalar operand (left to itself the
// compiler narrows EXEC around the computation of `a` instead: two scalar instructions where the scalar unit is scarce)
PZG_FN uint32_t mask_select(uint64_t m, uint32_t k, uint32_t a, uint32_t b)
{
#if PZG_DEVICE_PASS
    (void)k;
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
#else
    return ((m >> k) & 1ull) ? a : b;
#endif
}

// bit k of the wave-uniform mask m ? a : 0 (the zero rides in the instruction: no register holds it)
PZG_FN uint32_t mask_keep(uint64_t m, uint32_t k, uint32_t a)
{
#if PZG_DEVICE_PASS
    (void)k;
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
    return r;
#else
    return ((m >> k) & 1ull) ? a : 0u;
#endif
}

// m1 ? a1 : m0 ? a0 : b -- two selects in ONE statement, so that both a0 and a1 (crossbar results, typically) are asked
// for before either is waited for
PZG_FN uint32_t mask_select2(uint64_t m0, uint64_t m1, uint32_t k, uint32_t a0, uint32_t a1, uint32_t b)
{
#if PZG_DEVICE_PASS
    (void)k;
    uint32_t r = b;
    asm("v_cndmask_b32_e64 %0, %0, %1, %3\n\tv_cndmask_b32_e64 %0, %0, %2, %4" : "+v"(r) : "v"(a0), "v"(a1), "s"(m0), "s"(m1));
    return r;
#else
    return ((m1 >> k) & 1ull) ? a1 : ((m0 >> k) & 1ull) ? a0 : b;
#endif
}

// number of set bits of m below lane k
PZG_FN uint32_t mbcnt_k(uint64_t m, uint32_t k)
{
#if PZG_DEVICE_PASS
    (void)k;
    return mbcnt(m);
#else
    return (uint32_t)__builtin_popcountll(m & ((1ull << k) - 1ull));
#endif
}

// 4 * (base + number of set bits of m below lane k), base4 = 4 * base: a lane index as the crossbar instructions address
// it (the shift and the addition are one v_lshl_add_u32)
PZG_FN uint32_t mbcnt_slot4_k(uint64_t m, uint32_t base4, uint32_t k)
{
#if PZG_DEVICE_PASS
    (void)k;
    return (mbcnt(m) << 2) + base4;
#else
    return base4 + 4u * (uint32_t)__builtin_popcountll(m & ((1ull << k) - 1ull));
#endif
}

// `width` ones from bit `off` on (ONE s_bfm_b64; width < 64, off < 64)
PZG_FN uint64_t bit_field_mask(uint32_t width, uint32_t off)
{
#if PZG_DEVICE_PASS
    uint64_t m;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(width), "s"(off));
    return m;
#else
    return ((1ull << width) - 1ull) << off;
#endif
}

// (hi:lo) >> r, r in [0,32): v_alignbit_b32
// the same on wave-uniform operands: stays a scalar 64-bit shift
PZG_FN uint32_t funnel_uniform(uint32_t hi, uint32_t lo, uint32_t r)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (r & 31u));
}

PZG_FN uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t r)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_alignbit(hi, lo, r);  // (only r[4:0] counts)
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (r & 31u));
#endif
}

// (x >> off[4:0]) & ((1 << width[4:0]) - 1): v_bfe_u32 with both field operands in registers (only their low five bits count)
PZG_FN uint32_t ubfe(uint32_t x, uint32_t off, uint32_t width)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_ubfe(x, off, width);
#else
    off &= 31u;
    width &= 31u;
    return (x >> off) & ((1u << width) - 1u);
#endif
}

// (a & 0xff) + (b & 0xff) as ONE instruction (SDWA: both operands' low bytes selected in the add itself)
PZG_FN uint32_t byte0_sum(uint32_t a, uint32_t b)
{
#if PZG_DEVICE_PASS
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return (a & 0xffu) + (b & 0xffu);
#endif
}

// (x << 16) + y as ONE instruction (v_lshl_add_u32)
PZG_FN uint32_t shl16_add(uint32_t x, uint32_t y)
{
#if PZG_DEVICE_PASS
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#else
    return (x << 16) + y;
#endif
}

// bytes {a.3, a.2, b.3, b.2}: the high halves of a and b side by side (v_perm_b32)
PZG_FN uint32_t hi_halves(uint32_t a, uint32_t b)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_perm(a, b, 0x07060302u);
#else
    return (a & 0xffff0000u) | (b >> 16);
#endif
}

// ---- cross-lane operations on whole LaneVecs (64 lanes) ----------------------------------------

// inclusive prefix sum across the 64 lanes
PZG_FN void lanes_iscan_add(LaneVec<uint32_t> &x)
{
#if PZG_DEVICE_PASS
    uint32_t v = x.v;
    v += dpp_zero<0x111, 0xf>(v);
    v += dpp_zero<0x112, 0xf>(v);
    v += dpp_zero<0x114, 0xf>(v);
    v += dpp_zero<0x118, 0xf>(v);
    v += dpp_zero<0x142, 0xa>(v);  // lane 15 of rows 0,2 into rows 1,3
    v += dpp_zero<0x143, 0xc>(v);  // lane 31 into rows 2,3
    x.v = v;
#else
    for (uint32_t k = 1; k < 64u; ++k) x.v[k] += x.v[k - 1];
#endif
}

// inclusive prefix maximum across the 64 lanes (values are small non-negative integers)
PZG_FN void lanes_iscan_max(LaneVec<uint32_t> &x)
{
#if PZG_DEVICE_PASS
    uint32_t v = x.v, t;
    t = dpp_zero<0x111, 0xf>(v); v = v > t ? v : t;
    t = dpp_zero<0x112, 0xf>(v); v = v > t ? v : t;
    t = dpp_zero<0x114, 0xf>(v); v = v > t ? v : t;
    t = dpp_zero<0x118, 0xf>(v); v = v > t ? v : t;
    t = dpp_zero<0x142, 0xa>(v); v = v > t ? v : t;
    t = dpp_zero<0x143, 0xc>(v); v = v > t ? v : t;
    x.v = v;
#else
    for (uint32_t k = 1; k < 64u; ++k) x.v[k] = x.v[k] > x.v[k - 1] ? x.v[k] : x.v[k - 1];
#endif
}

// out[k] = src[idx[k] & 63]  (ds_bpermute_b32: a gather through the LDS crossbar, no LDS memory)
PZG_FN void lanes_gather(LaneVec<uint32_t> &out, const LaneVec<uint32_t> &src, const LaneVec<uint32_t> &idx)
{
#if PZG_DEVICE_PASS
    out.v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx.v << 2), (int)src.v);
#else
    LaneVec<uint32_t> tmp;
    for (uint32_t k = 0; k < 64u; ++k) tmp.v[k] = src.v[idx.v[k] & 63u];
    out = tmp;
#endif
}

// out[dst[k] & 63] = src[k]  (ds_permute_b32: a scatter through the LDS crossbar); lanes nobody
// sends to receive 0.  Callers send colliding lanes only to a lane whose result they ignore.
PZG_FN void lanes_scatter(LaneVec<uint32_t> &out, const LaneVec<uint32_t> &src, const LaneVec<uint32_t> &dst)
{
#if PZG_DEVICE_PASS
    out.v = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst.v << 2), (int)src.v);
#else
    LaneVec<uint32_t> tmp;
    for (uint32_t k = 0; k < 64u; ++k) tmp.v[k] = 0u;
    for (uint32_t k = 0; k < 64u; ++k) tmp.v[dst.v[k] & 63u] = src.v[k];
    out = tmp;
#endif
}

// the same with the destinations given as 4 * lane (the crossbar's own addressing)
PZG_FN void lanes_scatter4(LaneVec<uint32_t> &out, const LaneVec<uint32_t> &src, const LaneVec<uint32_t> &dst4)
{
#if PZG_DEVICE_PASS
    out.v = (uint32_t)__builtin_amdgcn_ds_permute((int)dst4.v, (int)src.v);
#else
    LaneVec<uint32_t> tmp;
    for (uint32_t k = 0; k < 64u; ++k) tmp.v[k] = 0u;
    for (uint32_t k = 0; k < 64u; ++k) tmp.v[(dst4.v[k] >> 2) & 63u] = src.v[k];
    out = tmp;
#endif
}

// mask of lanes whose predicate is set (the LaneVec holds 0/1)
PZG_FN uint64_t lanes_ballot(const LaneVec<uint32_t> &p)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_ballot_w64(p.v != 0u);
#else
    uint64_t m = 0;
    for (uint32_t k = 0; k < 64u; ++k) m |= (uint64_t)(p.v[k] != 0u) << k;
    return m;
#endif
}

// the same for a LaneVec of predicates (no 0/1 integers in between)
PZG_FN uint64_t lanes_ballot(const LaneVec<bool> &p)
{
#if PZG_DEVICE_PASS
    return __builtin_amdgcn_ballot_w64(p.v);
#else
    uint64_t m = 0;
    for (uint32_t k = 0; k < 64u; ++k) m |= (uint64_t)p.v[k] << k;
    return m;
#endif
}

PZG_FN uint32_t ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
PZG_FN uint32_t clz64(uint64_t m) { return (uint32_t)__builtin_clzll(m); }

PZG_FN uint32_t bitrev32(uint32_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    return __builtin_bswap32(x);
#endif
}

} // namespace pzg
