// pzg_inflate_kernel.h -- inflate_kernel<RING_BITS, FIXUP, GZIP>, the persistent stream-wave kernel around Decoder::run()
// (inflate_core.h), for the two translation units that instantiate it: pzg_kernels.hip (the zlib instances) and
// pzg_kernels_b.hip (the gzip instances and the resumable decoder's kernel, compiled without the SDWA peephole -- see the Makefile).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "inflate_core.h"
#include "pzg_launch.h"

namespace pzg {

// ------------------------------------------------------------------------------------------------
// inflate: a persistent grid of one-wave workgroups (block = 64 threads), each pulling stream indices
// from a device counter.  LDS per workgroup = sizeof(WaveLds): 36 KiB at RING_BITS = 15 (four
// stream-waves per CU, one per SIMD) down to 6 KiB at RING_BITS = 11 (26 per CU).
// Waves per SIMD each instance is compiled for (its VGPR budget: 512 / waves, in steps of 8) and the
// resident stream-waves per CU that follow from it and from sizeof(WaveLds) against the 160 KiB of LDS.
#ifndef PZG_MIN_WAVES_11
#define PZG_MIN_WAVES_11 7
#endif
// (the gzip instance of ring 11: rounds 3-4 96 vector registers, five waves per SIMD; round 5 80, six waves; compiled without the
// SDWA peephole -- whose operands must be registers: a dozen small constants held in vector registers for the kernel's whole
// life -- it needs 70: seven waves like the zlib instance, 26 stream-waves per CU, nothing in scratch)
#ifndef PZG_MIN_WAVES_11_GZIP
#define PZG_MIN_WAVES_11_GZIP 7
#endif
constexpr int waves_per_simd(int ring_bits, bool gzip = false)
{
    return ring_bits <= 11 ? (gzip ? PZG_MIN_WAVES_11_GZIP : PZG_MIN_WAVES_11) : ring_bits == 12 ? 5 : ring_bits == 13 ? 4 : ring_bits == 14 ? 2 : 1;
}
template <int RING_BITS, bool GZIP = false>
constexpr uint32_t waves_per_cu()
{
    constexpr uint32_t by_lds = (160u * 1024u) / (uint32_t)((sizeof(WaveLds<RING_BITS>) + 511u) / 512u * 512u);
    constexpr uint32_t by_vgpr = 4u * (uint32_t)(RING_BITS >= 13 ? 4 : waves_per_simd(RING_BITS, GZIP));  // rings 13-15 fit 128 VGPRs
    return by_lds < by_vgpr ? by_lds : by_vgpr;
}

// The launch arguments are NOT kept in scalar registers across a stream: the decoder's own wave-uniform state
// already fills the scalar register file (what does not fit is spilled into vector-register lanes, and uniform
// values start living in vector registers inside the hot loop).  They are read from the kernel-argument segment
// when a stream starts and again when its results are stored; the empty asm makes the pointer opaque at those two
// points so that the loads are not hoisted out of the stream loop.
typedef const InflateArgs __attribute__((address_space(4))) *LaunchArgs;  // (constant address space: scalar loads)
__device__ __forceinline__ LaunchArgs launch_args()
{
    LaunchArgs kp = (LaunchArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp)::"memory");
    return kp;
}

#if defined(PZG_LAB) && defined(PZG_LAB_SEED_PROFILE)
// Lab builds only (tests/test_gpu_parity.py: the persistent profile, fuzzed on the device -- VERDICT r5 item 5): before stream i the
// wave's profile words are overwritten with a well-marked profile that no stream ever taught it -- random, unsorted, decreasing,
// constant, squeezed or stretched quantiles, odd extents and token counts (one stream in eight keeps what the wave learnt).
__device__ __forceinline__ void lab_seed_profile(uint32_t *prof, uint32_t i)
{
    const uint32_t lane = threadIdx.x, kind = i & 7u;
    uint64_t z = ((uint64_t)i << 8 | lane) * 0x9E3779B97F4A7C15ull + 0x5EED0006ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const uint32_t r = (uint32_t)(z >> 32);
    if (kind == 7u) return;
    uint32_t q = r;
    if (kind == 1u) q = r & 0x3ffffu;
    else if (kind == 2u) q = lane * ((i >> 3) % 4000u + 1u);
    else if (kind == 3u) q = (63u - lane) * 1000u;
    else if (kind == 4u) q = (i >> 3) % 3u == 0u ? 0u : 4095u;
    else if (kind == 5u) q = lane == (i >> 3) % 63u + 1u ? (i & 8u ? 0u : 0xffffffffu) : lane * 1000u;
    else if (kind == 6u) q = 0u;
    prof[lane] = q;
    if (lane < 5u) {
        const uint32_t q63 = kind == 2u ? 63u * ((i >> 3) % 4000u + 1u) : kind == 5u ? 63000u : kind == 3u || kind == 6u ? 0u : kind == 4u ? q : 0x3ffffu;
        const uint32_t xs[4] = {q63 + 1u, 1u << 18, q63 + 5000u, q63}, ts[4] = {64u, 5000u, 64u * 192u, 64u * 192u + 1u};
        const uint32_t w = lane == 0u ? xs[(i >> 3) & 3u] : lane == 1u ? ts[(i >> 5) & 3u] : lane == 2u ? 0x51DF0A7Eu : 0u;
        prof[64u + lane] = kind == 0u ? r : w;
    }
}
#endif

template <int RING_BITS, bool FIXUP, bool GZIP = false>
__global__ __launch_bounds__(64, waves_per_simd(RING_BITS, GZIP)) void inflate_kernel(InflateArgs)
{
    __shared__ WaveLds<RING_BITS> lds;
    if (FIXUP && !GZIP && blockIdx.x == 0 && threadIdx.x == 0 && launch_args()->bundle_report)  // (the launch's last kernel: see pzg_api.cpp launch_device)
        *launch_args()->bundle_report = 1u + __hip_atomic_load(launch_args()->counter + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (FIXUP && __builtin_nontemporal_load(launch_args()->counter + 1) == 0u) return;  // nothing was handed back
    // (bundles, pzg_bundle_kernel.h: the launch's small streams of the fixed code are done; word 3 counts what was left to this kernel)
    // (... and word 4 what they decoded: with nothing decoded -- a batch of other streams -- no stream's status is looked at below)
    bool filter = false;
    if (!FIXUP && !GZIP && launch_args()->bundle != 0u) {
        // (uni(): a value that comes out of a vector load is lane-dependent to the compiler, and ONE lane-dependent branch in the
        // stream loop sends the whole loop through its structurizer -- see inflate_core.h hot_loop)
        // (device-scope loads, served by the L2: as non-temporal loads -- 6,656 waves asking one memory channel for the same word
        // twice -- they cost the launch 130 us, measured)
        if (uni(__hip_atomic_load(launch_args()->counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) return;
        filter = uni(__hip_atomic_load(launch_args()->counter + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u;
    }
    if (threadIdx.x == 0) lds.fixed_ready = 0u;  // LDS is not zeroed at launch
    __syncthreads();
    // Persistent stream-waves: the grid is sized to the residency of the chip and every wave pulls
    // stream indices from one device-scope counter until the batch is drained (a returning atomic is
    // ~0.3-1 us, nothing next to a >= 50 us stream; launching one workgroup per stream instead costs
    // more in dispatch than the small streams take to decode).
    for (;;) {
        uint32_t i = 0;
        StreamResult r;
        {
            LaunchArgs a = launch_args();
            if (threadIdx.x == 0) i = atomicAdd(a->counter, 1u);
            i = uni(i);
            if (i >= a->n) break;
#if !defined(PZG_PROFILE)
            if (a->order) i = a->order[i];
#endif
            // FIXUP pass (32 KiB ring): only the streams a small-ring launch handed back
            if (FIXUP && a->status[i] != ST_RETRY_FULL_RING) continue;
            if (!FIXUP && !GZIP && filter) {  // a bundle's stream: done
                asm volatile("" ::: "memory");  // (or the load is hoisted in front of the test: a trip to memory per stream, measured)
                if (uni((uint32_t)a->status[i]) != (uint32_t)ST_BUNDLE_TODO) continue;
            }
            Decoder<RING_BITS, GZIP> dec(lds);
            if (!FIXUP && a->strip && blockIdx.x < a->strip_waves) dec.strip = a->strip + (size_t)blockIdx.x * Decoder<RING_BITS, GZIP>::STRIP_WORDS;
            const uint8_t *dict = nullptr;
            uint32_t dict_len = 0;
            if (!GZIP && a->dict_len) {  // extension (PZG_FDICT): this stream's preset dictionary, if it has one
                const uint64_t dl = a->dict_len[i];
                dict = a->dict_base + a->dict_off[i];
                dict_len = dl > 0xffffffffull ? 0xffffffffu : (uint32_t)dl;
            }
#if defined(PZG_LAB) && defined(PZG_LAB_SEED_PROFILE)
            if (dec.strip) {
                static_assert(Decoder<RING_BITS, GZIP>::PROF_MAGIC == 0x51DF0A7Eu, "lab_seed_profile marks its profiles with the kernel's word");
                lab_seed_profile(dec.strip + Decoder<RING_BITS, GZIP>::PROF_OFF, i);
                __syncthreads();
            }
#endif
            dec.run(a->in_base + a->in_off[i], a->in_len[i], a->out_base + a->out_off[i], a->out_cap[i], &r, dict, dict_len);
#if defined(PZG_PROFILE)
            // diagnostic build: the 16 phase counters of stream i go to prof_out[16*i ..]
            if (threadIdx.x == 0 && a->prof_out)
                for (int q = 0; q < 16; ++q) a->prof_out[16 * (size_t)i + q] = dec.prof[q];
#endif
        }
        LaunchArgs a = launch_args();
        if (threadIdx.x == 0) {
            a->status[i] = r.status;
            if (!FIXUP && r.status == ST_RETRY_FULL_RING) atomicAdd(a->counter + 1, 1u);
            a->out_len[i] = r.out_len;
            if (a->detail) {
                a->detail[2 * (size_t)i] = r.detail0;
                a->detail[2 * (size_t)i + 1] = r.detail1;
            }
            if (a->in_used) a->in_used[i] = r.in_used;
            if (a->adler) a->adler[i] = GZIP ? 0u : r.adler;  // gzip: crc32_verify_kernel fills in the CRC-32
            if (GZIP) {
                a->gz_expect[2 * (size_t)i] = r.gz_crc;
                a->gz_expect[2 * (size_t)i + 1] = (uint32_t)r.out_len;
            }
        }
        __syncthreads();
    }
}


}  // namespace pzg
