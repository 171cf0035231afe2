// pzg_kernels_b.hip -- the kernels that are compiled WITHOUT the compiler's SDWA peephole (Makefile: KERNELFLAGS_B):
//
//   inflate_resume_kernel           the resumable decoder (decompressIncremental)
//   inflate_kernel<RB, *, true>     the gzip instances
//
// SDWA forms save a vector instruction here and there (an extract folded into an add) but take their operands from registers
// only: with the peephole on, a dozen small constants live in vector registers from the kernel's first line to its last.  The
// zlib instances pay that and are ~1 % faster for it (measured, round 5: 301.6 vs 298.6 GiB/s on the headline batch).  These two
// have more state: without it the resumable kernel needs no scratch memory (12 spilled vector registers with it) and runs
// 27 % faster (31.1 vs 24.4 GiB/s, bench.py's incremental leg), and the ring-11 gzip instance fits 72 registers (80 and one spill).
#include "pzg_inflate_kernel.h"

namespace pzg {

hipError_t launch_inflate_gzip(const InflateArgs &a, int ring_bits, bool fixup, uint32_t waves, hipStream_t stream)
{
    dim3 grid(waves), block(64);
    if (fixup)
        hipLaunchKernelGGL((inflate_kernel<15, true, true>), grid, block, 0, stream, a);
    else if (ring_bits == 15)
        hipLaunchKernelGGL((inflate_kernel<15, false, true>), grid, block, 0, stream, a);
    else if (ring_bits == 14)
        hipLaunchKernelGGL((inflate_kernel<14, false, true>), grid, block, 0, stream, a);
    else if (ring_bits == 13)
        hipLaunchKernelGGL((inflate_kernel<13, false, true>), grid, block, 0, stream, a);
    else if (ring_bits == 12)
        hipLaunchKernelGGL((inflate_kernel<12, false, true>), grid, block, 0, stream, a);
    else if (ring_bits == 11)
        hipLaunchKernelGGL((inflate_kernel<11, false, true>), grid, block, 0, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The resumable decoder (decompressIncremental, Monad.hs:163-197): one launch continues a batch of suspended decoders,
// one wave each, as far as their new input and output room go.  Round 4: the small-ring instance (PZG_RES_RING = 12: 8 KiB
// of LDS, four waves per SIMD, 16 decoders per CU where the 32 KiB LDS ring allowed 4).  What is older than the ring comes
// from the decoder's own 32 KiB history in HBM (every flush writes there as well as to the call's room), and a call saves /
// restores 8 KiB of LDS image instead of 37.
#ifndef PZG_RES_RING
#define PZG_RES_RING 12
#endif
#ifndef PZG_RES_WAVES_PER_SIMD
#define PZG_RES_WAVES_PER_SIMD 4
#endif
constexpr int RES_RING = PZG_RES_RING;
__global__ __launch_bounds__(64, RES_RING == 15 ? 1 : PZG_RES_WAVES_PER_SIMD) void inflate_resume_kernel(ResumeArgs a)
{
    __shared__ WaveLds<RES_RING> lds;
    for (;;) {
        uint32_t i = 0;
        if (threadIdx.x == 0) i = atomicAdd(a.counter, 1u);
        i = uni(i);
        if (i >= a.n) break;
        uint8_t *slot = a.state_base + (size_t)i * a.state_stride;
        ResumeState *rs = (ResumeState *)slot;
        uint32_t *image = (uint32_t *)(slot + ResumeSlot<RES_RING>::IMAGE_OFF);
        Decoder<RES_RING, false, true> dec(lds);
        if (a.strip && blockIdx.x < a.strip_waves) dec.strip = a.strip + (size_t)blockIdx.x * Decoder<RES_RING, false, true>::STRIP_WORDS;
        StreamResult r;
        uint32_t chunks = 0;
        dec.run_resume(rs, image, slot + ResumeSlot<RES_RING>::HIST_OFF, a.in_base + a.in_off[i], a.in_len[i], a.out_base + a.out_off[i],
                       a.out_cap[i], a.final_in ? (uint32_t)a.final_in[i] : 0u, &r, &chunks);
        if (threadIdx.x == 0) {
            a.status[i] = r.status;
            a.out_len[i] = r.out_len;
            a.in_used[i] = r.in_used;
            a.chunks[i] = chunks;
            if (a.adler) a.adler[i] = r.adler;
            if (a.detail) {
                a.detail[2 * (size_t)i] = r.detail0;
                a.detail[2 * (size_t)i + 1] = r.detail1;
            }
        }
        if (a.dense) {
            // what this decoder delivered, once more, behind what the others of its range delivered: the host then fetches ONE
            // linear span per range instead of rooms that are mostly empty (the bytes are this wave's own stores of a moment
            // ago: L2 hits; 64 lanes x 16 bytes per step, four steps in flight)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const uint32_t nv = (uint32_t)((r.out_len + 15u) >> 4);
            uint32_t at = 0;
            if (threadIdx.x == 0) at = atomicAdd(a.dense_cursor, nv);
            at = uni(at);
            const uint64_t off = a.dense_region + 16ull * at;
            if (threadIdx.x == 0) a.dense_off[i] = off;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the flushes' stores have landed
            const u32x4 *src = (const u32x4 *)(const void *)(a.out_base + a.out_off[i]);
            u32x4 *dst = (u32x4 *)(void *)(a.dense + off);
            for (uint32_t v0 = 0; v0 < nv; v0 += 256u) {
                u32x4 t[4];
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) {
                    const uint32_t v = v0 + 64u * q + threadIdx.x;
                    t[q] = __builtin_nontemporal_load(src + (v < nv ? v : nv - 1u));
                }
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) {
                    const uint32_t v = v0 + 64u * q + threadIdx.x;
                    if (v < nv) dst[v] = t[q];
                }
            }
        }
        __syncthreads();
    }
}

size_t resume_scalar_bytes() { return sizeof(ResumeState); }
size_t resume_state_bytes() { return ResumeSlot<RES_RING>::BYTES; }

size_t resume_strip_wave_bytes() { return (size_t)Decoder<RES_RING, false, true>::STRIP_WORDS * sizeof(uint32_t); }
uint32_t resume_launch_waves(int num_cus, uint32_t n)
{
    constexpr uint32_t by_lds = (160u * 1024u) / (uint32_t)((sizeof(WaveLds<RES_RING>) + 511u) / 512u * 512u);
    constexpr uint32_t by_vgpr = RES_RING == 15 ? 4u : 4u * PZG_RES_WAVES_PER_SIMD;
    const uint32_t waves = (uint32_t)num_cus * (by_lds < by_vgpr ? by_lds : by_vgpr);
    return waves > n ? n : waves;
}

hipError_t launch_resume(const ResumeArgs &a, int num_cus, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(a.counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(inflate_resume_kernel, dim3(resume_launch_waves(num_cus, a.n)), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace pzg
