// pzg_kernels.hip -- gfx950 kernels + their launchers.
//
//   inflate_kernel<RING_BITS>   one workgroup = one wavefront = one zlib stream (inflate_core.h)
//   adler32_partial_kernel      Adler32.hs as an HBM-bound streaming reduction (BASELINE config 2)
//   adler32_combine_kernel      folds the per-wave partials in order
//
// Written for CDNA4 only: wave64, LDS ring per wave, no MFMA (there is no contraction on this path).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pzg_inflate_kernel.h"
#include "pzg_bundle_kernel.h"

namespace pzg {

// ------------------------------------------------------------------------------------------------
// CRC-32 (RFC 1952 section 8) of each decoded gzip member against its trailer: an extension (the reference has
// no gzip).  One wave per stream.  The stream is cut into 64 equal slices by padding
// it at the FRONT with zero bytes (a zero register stays zero over zero bytes, so the padding is free);
// lane l runs slice l through four byte-indexed tables in LDS (slicing-by-4), starting from 0xffffffff
// in the lane that holds the first real byte and from 0 in the others; the 64 registers are then folded
// pairwise, R_left * x^(8 * slice * 2^level) + R_right over GF(2) modulo the CRC polynomial.
__host__ __device__ constexpr uint32_t gf2_mulmod_c(uint32_t a, uint32_t b)  // reflected bit order, poly 0xedb88320
{
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (0xedb88320u & (0u - (b & 1u)));
    }
    return p;
}
__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma nounroll
    for (int i = 0; i < 32; ++i) {
        p ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = (b >> 1) ^ (0xedb88320u & (0u - (b & 1u)));
    }
    return p;
}
// x^(8 * 2^j) mod P for j = 0 .. 47: the factor that moves a CRC register over 2^j zero bytes.  Constants, so the
// power for any span is a product over the set bits of its length -- no squarings at run time (a power of two: one
// table entry, no multiplication at all).
struct CrcPowers {
    uint32_t c[48];
    constexpr CrcPowers() : c{}
    {
        uint32_t x = 0x00800000u;  // x^8 (bit 31 is x^0)
        for (int j = 0; j < 48; ++j) {
            c[j] = x;
            x = gf2_mulmod_c(x, x);
        }
    }
};
__constant__ const CrcPowers crc_powers{};
// x^(8 * n * 2^shift) mod P, n and shift wave-uniform (n * 2^shift < 2^48)
__device__ __forceinline__ uint32_t crc_zero_bytes_factor(uint64_t n, uint32_t shift)
{
    uint32_t pw = 0x80000000u;  // x^0
    bool first = true;
#pragma nounroll
    for (uint32_t j = 0; n != 0; ++j, n >>= 1) {
        if (n & 1u) {
            const uint32_t c = crc_powers.c[j + shift];
            pw = first ? c : gf2_mulmod(pw, c);
            first = false;
        }
    }
    return pw;
}

// One wave per stream, four waves (streams) per workgroup sharing the tables.  The stream, padded at the FRONT with
// zero bytes to a multiple of 1 KiB (a zero register stays zero over zero bytes), is read in 1 KiB blocks, lane l
// taking bytes [16 l, 16 l + 16) of every block: one coalesced 1 KiB wave-load per block (an earlier version gave each
// lane one contiguous slice: 64 cache lines per load instruction, 0.95 TB/s).  CRCs are linear over GF(2): with a
// zero initial register, lane l's accumulator advances over the 1008 bytes of the other lanes (four lookups in the
// tables ADV) and then over its own 16 (slicing-by-4, sixteen lookups in T); at the end the accumulators are moved
// over the 16 (63 - l) bytes behind each lane's last chunk and XORed together, and the contribution of the real
// initial register 0xffffffff -- advanced over the whole length -- is added.
constexpr uint32_t CRC_BLOCK = 1024u, CRC_WAVES = 4u;
__global__ __launch_bounds__(64 * CRC_WAVES) void crc32_verify_kernel(InflateArgs a)
{
    __shared__ uint32_t T[4][256], ADV[4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t v = tid; v < 256u; v += 64u * CRC_WAVES) {
        uint32_t r = v;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (0xedb88320u & (0u - (r & 1u)));
        T[0][v] = r;
    }
    __syncthreads();
    const uint32_t f_adv = crc_zero_bytes_factor(CRC_BLOCK - 16u, 0u);
    for (uint32_t v = tid; v < 256u; v += 64u * CRC_WAVES) {
        uint32_t r = T[0][v];
        for (int k = 1; k < 4; ++k) {
            r = (r >> 8) ^ T[0][r & 0xffu];
            T[k][v] = r;
        }
        for (int k = 0; k < 4; ++k) ADV[k][v] = gf2_mulmod(v << (8 * k), f_adv);
    }
    // x^(8 * 16 * (63 - lane)): what moves this lane's accumulator to the end of the stream
    uint32_t f_lane = 0x80000000u;
    {
        const uint32_t nb = 16u * (63u - lane);
#pragma nounroll
        for (uint32_t j = 0; j < 10u; ++j) {
            const uint32_t t = gf2_mulmod(f_lane, crc_powers.c[j]);
            f_lane = (nb >> j) & 1u ? t : f_lane;
        }
    }
    __syncthreads();
    for (uint32_t i = blockIdx.x * CRC_WAVES + wave; i < a.n; i += gridDim.x * CRC_WAVES) {
        // every member's ISIZE was checked as it was decoded; a mismatch is reported unless the CRC-32 is wrong as well
        // (zlib's order).  (Output larger than its capacity: PZG_E_OUT_TOO_SMALL, nothing stored to check.)
        if (a.status[i] != ST_OK && a.status[i] != ST_GZIP_ISIZE) continue;
        const uint64_t len = a.out_len[i];
        const uint8_t *p = a.out_base + a.out_off[i];
        const uint64_t pad = (CRC_BLOCK - (len & (CRC_BLOCK - 1u))) & (CRC_BLOCK - 1u);  // zero bytes in front
        const uint64_t nblk = (len + pad) / CRC_BLOCK;
        uint32_t acc = 0;
        for (uint64_t blk = 0; blk < nblk; ++blk) {
            const uint64_t pp = blk * CRC_BLOCK + 16u * lane;  // this lane's chunk in the padded stream
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (blk != 0u || pad == 0u) {  // (wave-uniform) the whole block is stream
                __builtin_memcpy(w, p + (pp - pad), 16);
            } else {  // the first block of a stream whose length is no multiple of 1 KiB: bytes in front of the stream are zero
                for (uint32_t t = 0; t < 16u; ++t) {
                    const uint64_t q = pp + t;
                    const uint32_t byte = q >= pad ? (uint32_t)p[q - pad] : 0u;
                    w[t >> 2] |= byte << (8u * (t & 3u));
                }
            }
            uint32_t reg = ADV[3][acc >> 24] ^ ADV[2][(acc >> 16) & 0xffu] ^ ADV[1][(acc >> 8) & 0xffu] ^ ADV[0][acc & 0xffu];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                reg ^= w[k];
                reg = T[3][reg & 0xffu] ^ T[2][(reg >> 8) & 0xffu] ^ T[1][(reg >> 16) & 0xffu] ^ T[0][reg >> 24];
            }
            acc = reg;
        }
        uint32_t reg = gf2_mulmod(acc, f_lane);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) reg ^= (uint32_t)__shfl_xor((int)reg, o, 64);
        if (lane == 0u) {
            reg ^= gf2_mulmod(0xffffffffu, crc_zero_bytes_factor(len, 0u));  // the initial register, advanced over the stream
            const uint32_t ours = len ? ~reg : 0u, theirs = a.gz_expect[2 * (size_t)i];
            if (a.adler) a.adler[i] = ours;
            if (theirs != ours) {
                a.status[i] = ST_CHECKSUM;
                if (a.detail) {
                    a.detail[2 * (size_t)i] = theirs;
                    a.detail[2 * (size_t)i + 1] = ours;
                }
            }
        }
    }
}

// stream-waves of a launch: the residency of the chip, or one per stream if there are fewer
static uint32_t launch_waves(int ring_bits, int num_cus, uint32_t n, uint32_t gzip)
{
    // Resident stream-waves per CU: 4 / 8 / 13 / 20 / 26 for rings 15 .. 11 (LDS-bound; ring 11: 72 VGPRs, 7 per SIMD by registers)
    const uint32_t per_cu = ring_bits == 15 ? waves_per_cu<15>() : ring_bits == 14 ? waves_per_cu<14>()
                            : ring_bits == 13 ? waves_per_cu<13>() : ring_bits == 12 ? waves_per_cu<12>()
                            : gzip ? waves_per_cu<11, true>() : waves_per_cu<11>();
    uint32_t waves = (uint32_t)num_cus * per_cu;
#if defined(PZG_LAB)   // lab builds only (tests/tools/exp_build.sh -DPZG_LAB): the residency sweep
    if (const char *e = getenv("PZG_WAVES")) waves = (uint32_t)atoi(e);
#endif
    return waves > n ? n : waves;
}
size_t inflate_strip_bytes(int ring_bits, int num_cus, uint32_t n, uint32_t gzip)
{
    return (size_t)launch_waves(ring_bits, num_cus, n, gzip) * Decoder<11>::STRIP_WORDS * sizeof(uint32_t);
}
size_t inflate_strip_wave_bytes() { return (size_t)Decoder<11>::STRIP_WORDS * sizeof(uint32_t); }

// PZG_OPT_PROFILE: word 67 of a wave's profile counts the streams for which it is not to be consulted (strip_profile_layout): all
// ones and a magic word -- "a profile is there, leave it alone for the next four billion streams" -- switches it off; zero switches
// it on again (what the wave learnt in between is still there: strip_profile_learn goes on writing the quantiles)
__global__ __launch_bounds__(256) void profile_switch_kernel(uint32_t *strip, uint32_t waves, uint32_t off)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= waves) return;
    uint32_t *p = strip + (size_t)w * Decoder<11>::STRIP_WORDS + Decoder<11>::PROF_OFF;
    if (off) {
        p[66] = Decoder<11>::PROF_MAGIC;
        p[67] = 0xffffffffu;
    } else {
        p[67] = 0u;
        p[68] = 0u;
    }
}
hipError_t launch_profile_switch(uint32_t *strip, uint32_t waves, bool off, hipStream_t stream)
{
    if (waves == 0u) return hipSuccess;
    hipLaunchKernelGGL(profile_switch_kernel, dim3((waves + 255u) / 256u), dim3(256), 0, stream, strip, waves, off ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_inflate(const InflateArgs &a_in, int ring_bits, int num_cus, hipStream_t stream)
{
    if (a_in.n == 0) return hipSuccess;
    InflateArgs a = a_in;
    if (a.gzip || a.dict_len) a.bundle = 0u;
    hipError_t e = hipMemsetAsync(a.counter, 0, 8 * sizeof(uint32_t), stream);  // [0] stream index, [1] streams handed back, [3] the bundles' (pzg_bundle_kernel.h)
    if (e != hipSuccess) return e;
    if (a.bundle) {  // the small streams of the fixed code first, 64 of the launch order to a wave
        hipLaunchKernelGGL(bundle_kernel, dim3((a.n + 63u) / 64u), dim3(64), 0, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
#if defined(PZG_LAB) && defined(PZG_LAB_BUNDLE_NOFLAG)  // (lab: what the bundle kernel's mere presence costs the kernel behind it)
        a.bundle = 0u;
#endif
    }
    const uint32_t waves = launch_waves(ring_bits, num_cus, a.n, a.gzip);
    dim3 grid(waves), block(64);
    // Ring size classes.  15: the whole 32 KiB DEFLATE window is an LDS ring (4 stream-waves per CU).
    // 12-14: a smaller near ring plus far back-references served from the stream's own flushed output
    // (more resident stream-waves per CU; the kernel is latency-bound, so that is what it scales with).
    // (the gzip instances live in pzg_kernels_b.hip)
#define PZG_LAUNCH_RING(RB)                                                                   \
    do {                                                                                      \
        if (a.gzip)                                                                           \
            e = launch_inflate_gzip(a, RB, false, waves, stream);                             \
        else                                                                                  \
            hipLaunchKernelGGL((inflate_kernel<RB, false, false>), grid, block, 0, stream, a); \
    } while (0)
    if (ring_bits == 15)
        PZG_LAUNCH_RING(15);
    else if (ring_bits == 14)
        PZG_LAUNCH_RING(14);
    else if (ring_bits == 13)
        PZG_LAUNCH_RING(13);
    else if (ring_bits == 12)
        PZG_LAUNCH_RING(12);
    else if (ring_bits == 11)
        PZG_LAUNCH_RING(11);
    else
        return hipErrorInvalidValue;
#undef PZG_LAUNCH_RING
    if (e != hipSuccess) return e;
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ring_bits != 15) {
        // streams whose output outgrew their capacity are redone by the pure-LDS-ring kernel
        e = hipMemsetAsync(a.counter, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        dim3 fgrid(a.n < 1024u ? a.n : 1024u);
        if (a.gzip) {
            e = launch_inflate_gzip(a, 15, true, fgrid.x, stream);
            if (e != hipSuccess) return e;
        } else
            hipLaunchKernelGGL((inflate_kernel<15, true, false>), fgrid, block, 0, stream, a);
    }
    e = hipGetLastError();
    if (e != hipSuccess || !a.gzip) return e;
    // gzip members: CRC-32 and ISIZE of every decoded stream against its trailer (one more pass over the output)
    const uint32_t cwg = (a.n + CRC_WAVES - 1u) / CRC_WAVES;
    dim3 cgrid(cwg < (uint32_t)num_cus * 8u ? cwg : (uint32_t)num_cus * 8u);
    hipLaunchKernelGGL(crc32_verify_kernel, cgrid, dim3(64 * CRC_WAVES), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Adler-32 (Adler32.hs:17-57) over one large buffer.
//
// The buffer is viewed as 16-byte vectors at aligned addresses; the bytes of the first and last
// vector that fall outside [buf, buf+len) are masked to zero (zeros add nothing to either sum and
// the position weights are taken from the END of the buffer, so leading padding is harmless).
// Each wave owns a contiguous run of 64 KiB blocks (4096 vectors); within a block lane l reads
// vector it*64 + l (one fully coalesced 1 KiB wave-load per iteration) and keeps three u32
// partial sums.  For a zero-initialised run X followed by Y:
//     SA = SA_x + SA_y          SB = SB_x + len_y * SA_x + SB_y          (mod 65521)
// with SA = sum d, SB = sum (n - pos) d.
constexpr uint32_t AD_BLOCK_VECS = 4096;  // 64 KiB per block: u32 lane sums cannot overflow
constexpr uint32_t AD_UNROLL = 8;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adler_acc(u32x4 v, uint32_t it, uint32_t &a, uint32_t &w, uint32_t &u)
{
    uint32_t s = sum4(v.x, sum4(v.y, sum4(v.z, sum4(v.w, 0u))));
    uint32_t t = dot4(v.x, 0x0D0E0F10u, dot4(v.y, 0x090A0B0Cu, dot4(v.z, 0x05060708u, dot4(v.w, 0x01020304u, 0u))));
    a += s;
    w += t;
    u += it * s;
}

__device__ __forceinline__ u32x4 mask_vec(u32x4 v, uint32_t lo, uint32_t hi)
{
    // keep bytes [lo, hi) of the 16
    uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            uint32_t idx = 4 * k + b;
            if (idx >= lo && idx < hi) m |= 0xffu << (8 * b);
        }
        x[k] &= m;
    }
    u32x4 r = {x[0], x[1], x[2], x[3]};
    return r;
}

__global__ __launch_bounds__(256) void adler32_partial_kernel(const uint8_t *abase, uint64_t nvec, uint32_t head_pad,
                                                             uint32_t tail_valid, uint64_t nblocks,
                                                             uint32_t blocks_per_wave, uint32_t *partials)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t b0 = wave * blocks_per_wave;
    uint64_t b1 = b0 + blocks_per_wave;
    if (b1 > nblocks) b1 = nblocks;
    uint32_t SA = 0, SB = 0;  // running sums of this wave's run, mod 65521
    uint64_t run_bytes = 0;
    const u32x4 *vbase = (const u32x4 *)abase;
    for (uint64_t b = b0; b < b1; ++b) {
        const uint64_t v0 = b * AD_BLOCK_VECS;
        const uint32_t nv = (uint32_t)((nvec - v0) < AD_BLOCK_VECS ? (nvec - v0) : AD_BLOCK_VECS);
        uint32_t a_l = 0, w_l = 0, u_l = 0;
        const bool edge = (b == 0) || (v0 + nv == nvec) || nv < AD_BLOCK_VECS;
        if (!edge) {
            const u32x4 *p = vbase + v0 + lane;
            for (uint32_t it = 0; it < AD_BLOCK_VECS / 64u; it += AD_UNROLL) {
                u32x4 v[AD_UNROLL];
#pragma unroll
                for (uint32_t q = 0; q < AD_UNROLL; ++q) v[q] = __builtin_nontemporal_load(p + (size_t)(it + q) * 64u);
#pragma unroll
                for (uint32_t q = 0; q < AD_UNROLL; ++q) adler_acc(v[q], it + q, a_l, w_l, u_l);
            }
        } else {
            for (uint32_t it = 0; it * 64u < nv; ++it) {
                const uint32_t j = it * 64u + lane;
                if (j < nv) {
                    u32x4 v = vbase[v0 + j];
                    const uint64_t gv = v0 + j;
                    uint32_t lo = gv == 0 ? head_pad : 0u;
                    uint32_t hi = gv == nvec - 1 ? tail_valid : 16u;
                    if (lo != 0u || hi != 16u) v = mask_vec(v, lo, hi);
                    adler_acc(v, it, a_l, w_l, u_l);
                }
            }
        }
        const uint32_t n = nv * 16u;  // block length in (padded) bytes
        int64_t bl = ((int64_t)n - 16 * (int64_t)lane - 16) * (int64_t)a_l + (int64_t)w_l - 1024 * (int64_t)u_l;
        uint32_t blk_b = wave_sum((uint32_t)((uint64_t)bl % ADLER_MOD));
        uint32_t blk_a = wave_sum(a_l) % ADLER_MOD;
        SB = (uint32_t)(((uint64_t)SB + (uint64_t)(n % ADLER_MOD) * SA + blk_b) % ADLER_MOD);
        SA = (SA + blk_a) % ADLER_MOD;
        run_bytes += n;
    }
    if (lane == 0) {
        partials[3 * wave + 0] = SA;
        partials[3 * wave + 1] = SB;
        partials[3 * wave + 2] = (uint32_t)(run_bytes % ADLER_MOD);
    }
}

// One wave folds the partials in order: lane l folds a contiguous slice, then lane 0 folds the 64 slices.
__global__ __launch_bounds__(64) void adler32_combine_kernel(const uint32_t *partials, uint32_t nwaves, uint32_t init,
                                                            uint64_t len, uint32_t tail_pad, uint32_t *out)
{
    __shared__ uint32_t sh[64 * 3];
    const uint32_t lane = threadIdx.x;
    const uint32_t per = (nwaves + 63u) / 64u;
    uint32_t SA = 0, SB = 0, N = 0;
    for (uint32_t k = lane * per; k < (lane + 1) * per && k < nwaves; ++k) {
        const uint32_t ya = partials[3 * k], yb = partials[3 * k + 1], yn = partials[3 * k + 2];
        SB = (uint32_t)(((uint64_t)SB + (uint64_t)yn * SA + yb) % ADLER_MOD);
        SA = (SA + ya) % ADLER_MOD;
        N = (N + yn) % ADLER_MOD;
    }
    sh[3 * lane] = SA;
    sh[3 * lane + 1] = SB;
    sh[3 * lane + 2] = N;
    __syncthreads();
    if (lane == 0) {
        SA = 0;
        SB = 0;
        for (uint32_t k = 0; k < 64; ++k) {
            const uint32_t ya = sh[3 * k], yb = sh[3 * k + 1], yn = sh[3 * k + 2];
            SB = (uint32_t)(((uint64_t)SB + (uint64_t)yn * SA + yb) % ADLER_MOD);
            SA = (SA + ya) % ADLER_MOD;
        }
        // the zero bytes padding the last vector moved the end of the buffer by tail_pad: every weight is
        // tail_pad too large, i.e. SB is tail_pad * SA too large
        SB = (uint32_t)(((uint64_t)SB + (uint64_t)ADLER_MOD * 16u - (uint64_t)tail_pad * SA) % ADLER_MOD);
        // start from `init` = (b0 << 16) | a0 :  A = a0 + SA ; B = b0 + len*a0 + SB   (Adler32.hs:22-27 unrolled)
        const uint32_t a0 = init & 0xffffu, b0 = init >> 16;
        const uint32_t A = (a0 + SA) % ADLER_MOD;
        const uint32_t B = (uint32_t)(((uint64_t)b0 + (len % ADLER_MOD) * a0 + SB) % ADLER_MOD);
        out[0] = (B << 16) | A;
    }
}

hipError_t launch_adler32(const uint8_t *buf, uint64_t len, uint32_t init, uint32_t *partials, uint32_t max_waves,
                          uint32_t *out, hipStream_t stream)
{
    const uint32_t head_pad = (uint32_t)((uintptr_t)buf & 15u);
    const uint8_t *abase = buf - head_pad;
    const uint64_t padded = head_pad + len;
    uint64_t nvec = (padded + 15u) >> 4;
    uint32_t tail_valid = (uint32_t)(padded & 15u);
    if (tail_valid == 0) tail_valid = 16;
    if (len == 0) nvec = 0;
    const uint64_t nblocks = (nvec + AD_BLOCK_VECS - 1) / AD_BLOCK_VECS;
    uint64_t nwaves = nblocks < max_waves ? nblocks : max_waves;
    if (nwaves == 0) nwaves = 1;
    const uint32_t bpw = (uint32_t)((nblocks + nwaves - 1) / nwaves);
    nwaves = nblocks ? (nblocks + bpw - 1) / bpw : 1;
    const uint32_t wgs = (uint32_t)((nwaves + 3) / 4);
    hipLaunchKernelGGL(adler32_partial_kernel, dim3(wgs), dim3(256), 0, stream, abase, nvec, head_pad, tail_valid,
                       nblocks, bpw ? bpw : 1u, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(adler32_combine_kernel, dim3(1), dim3(64), 0, stream, partials, wgs * 4u, init, len,
                       16u - tail_valid, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Adler-32 of MANY buffers (the batched form of BASELINE config 2: 262,144 x 64 KiB): one wave per buffer, the same
// 16-byte vectors, lane sums and block fold as adler32_partial_kernel; a persistent grid strides over the buffers.
__global__ __launch_bounds__(256) void adler32_many_kernel(const uint8_t *base, const uint64_t *off, const uint64_t *len, uint32_t *out,
                                                          uint32_t n)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t i = wave; i < n; i += nwaves) {
        const uint8_t *buf = base + off[i];
        const uint64_t blen = len[i];
        const uint32_t head_pad = (uint32_t)((uintptr_t)buf & 15u);
        const u32x4 *vbase = (const u32x4 *)(buf - head_pad);
        const uint64_t padded = head_pad + blen;
        const uint64_t nvec = blen ? (padded + 15u) >> 4 : 0u;
        uint32_t tail_valid = (uint32_t)(padded & 15u);
        if (tail_valid == 0) tail_valid = 16;
        uint32_t SA = 0, SB = 0;
        for (uint64_t v0 = 0; v0 < nvec; v0 += AD_BLOCK_VECS) {
            const uint32_t nv = (uint32_t)((nvec - v0) < AD_BLOCK_VECS ? (nvec - v0) : AD_BLOCK_VECS);
            uint32_t a_l = 0, w_l = 0, u_l = 0;
            const bool edge = v0 == 0 || v0 + nv == nvec;
            if (!edge) {
                const u32x4 *p = vbase + v0 + lane;
                for (uint32_t it = 0; it < AD_BLOCK_VECS / 64u; it += AD_UNROLL) {
                    u32x4 v[AD_UNROLL];
#pragma unroll
                    for (uint32_t q = 0; q < AD_UNROLL; ++q) v[q] = __builtin_nontemporal_load(p + (size_t)(it + q) * 64u);
#pragma unroll
                    for (uint32_t q = 0; q < AD_UNROLL; ++q) adler_acc(v[q], it + q, a_l, w_l, u_l);
                }
            } else {
                for (uint32_t it = 0; it * 64u < nv; ++it) {
                    const uint32_t j = it * 64u + lane;
                    if (j < nv) {
                        u32x4 v = __builtin_nontemporal_load(vbase + v0 + j);
                        const uint64_t gv = v0 + j;
                        const uint32_t lo = gv == 0 ? head_pad : 0u, hi = gv == nvec - 1 ? tail_valid : 16u;
                        if (lo != 0u || hi != 16u) v = mask_vec(v, lo, hi);
                        adler_acc(v, it, a_l, w_l, u_l);
                    }
                }
            }
            const uint32_t nb = nv * 16u;
            int64_t bl = ((int64_t)nb - 16 * (int64_t)lane - 16) * (int64_t)a_l + (int64_t)w_l - 1024 * (int64_t)u_l;
            const uint32_t blk_b = wave_sum((uint32_t)((uint64_t)bl % ADLER_MOD));
            const uint32_t blk_a = wave_sum(a_l) % ADLER_MOD;
            SB = (uint32_t)(((uint64_t)SB + (uint64_t)(nb % ADLER_MOD) * SA + blk_b) % ADLER_MOD);
            SA = (SA + blk_a) % ADLER_MOD;
        }
        if (lane == 0) {
            const uint32_t tail_pad = 16u - tail_valid;  // zero bytes behind the buffer: every weight is that much too large
            SB = (uint32_t)(((uint64_t)SB + (uint64_t)ADLER_MOD * 16u - (uint64_t)tail_pad * SA) % ADLER_MOD);
            const uint32_t A = (1u + SA) % ADLER_MOD;                                   // Adler32.hs:22-27 from the initial (1, 0)
            const uint32_t B = (uint32_t)(((blen % ADLER_MOD) + SB) % ADLER_MOD);
            out[i] = blen ? (B << 16) | A : 1u;
        }
    }
}

hipError_t launch_adler32_many(const uint8_t *base, const uint64_t *off, const uint64_t *len, uint32_t *out, uint32_t n, int num_cus,
                               hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint32_t wgs = (uint32_t)num_cus * 8u;  // 32 waves per CU
    if (wgs > (n + 3u) / 4u) wgs = (n + 3u) / 4u;
    hipLaunchKernelGGL(adler32_many_kernel, dim3(wgs), dim3(256), 0, stream, base, off, len, out, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Launch order for the persistent stream-waves: longest streams first (by capacity), so the tail of a mixed batch is
// filled by short streams instead of waiting for one long one.  A counting sort over the 64 power-of-two size
// classes, descending; inside a class the order is whatever the atomics give (the sizes are within 2x).
__global__ __launch_bounds__(256) void order_hist_kernel(const uint64_t *out_cap, uint32_t n, uint32_t *hist)
{
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t c = out_cap[i];
        atomicAdd(&h[c ? (uint32_t)__builtin_clzll(c) : 63u], 1u);  // class 0 = the largest
    }
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(64) void order_scan_kernel(uint32_t *hist)  // hist[0..64) -> cursors in hist[64..128)
{
    uint32_t v = hist[threadIdx.x], incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64);
        if ((int)threadIdx.x >= o) incl += y;
    }
    hist[64 + threadIdx.x] = incl - v;
}
__global__ __launch_bounds__(256) void order_scatter_kernel(const uint64_t *out_cap, uint32_t n, uint32_t *hist, uint32_t *order)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t c = out_cap[i];
        order[atomicAdd(&hist[64u + (c ? (uint32_t)__builtin_clzll(c) : 63u)], 1u)] = i;
    }
}

hipError_t launch_order(const uint64_t *out_cap, uint32_t n, uint32_t *order, uint32_t *scratch, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(scratch, 0, 128 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t wgs = n < 256u * 1024u ? (n + 255u) / 256u : 1024u;
    hipLaunchKernelGGL(order_hist_kernel, dim3(wgs), dim3(256), 0, stream, out_cap, n, scratch);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(64), 0, stream, scratch);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(wgs), dim3(256), 0, stream, out_cap, n, scratch, order);
    return hipGetLastError();
}

}  // namespace pzg
