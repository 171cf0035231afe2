// pzg_bundle_kernel.h -- bundle_kernel: the launch's small streams of the fixed code (Deflate.hs:79-82), 64 to a wave, one lane per
// stream (bundle_core.h).  One workgroup = one wave = 64 consecutive streams of the launch order; what a lane does not take --
// every stream that is not plain -- gets status ST_BUNDLE_TODO and is decoded by the ordinary kernel that follows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bundle_core.h"
#include "pzg_launch.h"

namespace pzg {

// counter words of a launch (InflateArgs::counter): [0] the ordinary kernel's stream index, [1] streams handed back to the
// fixup pass, [3] streams the bundles left to the ordinary kernel, [4] streams they decoded; [16..19]: what the bundles' idle lanes read
enum : uint32_t { CTR_TODO = 3, CTR_CLEAN = 4, CTR_COMMON = 16 };

__global__ __launch_bounds__(64, 1) void bundle_kernel(InflateArgs a)
{
    __shared__ BundleLds lds;
    const uint32_t lane = threadIdx.x, pos = blockIdx.x * 64u + lane;
    const bool have = pos < a.n;
    const uint32_t i = have ? (a.order ? a.order[pos] : pos) : 0u;
    const uint64_t len = a.in_len[i], cap = a.out_cap[i];
    Bundle::In bi;
    Bundle::Out bo;
    const bool on = have && len >= 8u && len < Bundle::MAX_BYTES && cap < Bundle::MAX_BYTES;
    const uint8_t *in = a.in_base + a.in_off[i];
    // (the first block's type, before anything is set up: a bundle of streams of another kind costs a few loads)
    uint32_t b2 = 0u;
    if (on) b2 = in[2];
    if (__builtin_amdgcn_ballot_w64(on && ((b2 >> 1) & 3u) == 1u) == 0ull) {
        if (have) a.status[i] = ST_BUNDLE_TODO;
        if (lane == 0u) atomicAdd(a.counter + CTR_TODO, have ? (a.n - pos < 64u ? a.n - pos : 64u) : 0u);
        return;
    }
    PZG_LV(bi.IN, lane) = in;
    PZG_LV(bi.OUT, lane) = a.out_base + a.out_off[i];
    PZG_LV(bi.LEN, lane) = on ? (uint32_t)len : 0u;
    PZG_LV(bi.CAP, lane) = on ? (uint32_t)cap : 0u;
    PZG_LV(bi.ON, lane) = on ? 1u : 0u;
    Bundle::run(lds, bi, a.counter + CTR_COMMON, bo);
    const bool clean = PZG_LV(bo.STATE, lane) == Bundle::BS_CLEAN;
    if (have) {
        if (clean) {
            a.status[i] = (int32_t)PZG_LV(bo.STATUS, lane);
            a.out_len[i] = PZG_LV(bo.OLEN, lane);
            if (a.detail) {
                a.detail[2 * (size_t)i] = PZG_LV(bo.D0, lane);
                a.detail[2 * (size_t)i + 1] = PZG_LV(bo.D1, lane);
            }
            if (a.in_used) a.in_used[i] = PZG_LV(bo.USED, lane);
            if (a.adler) a.adler[i] = PZG_LV(bo.ADLER, lane);
        } else {
            a.status[i] = ST_BUNDLE_TODO;
        }
    }
    const uint64_t todo = __builtin_amdgcn_ballot_w64(have && !clean), done = __builtin_amdgcn_ballot_w64(have && clean);
    if (lane == 0u && todo != 0ull) atomicAdd(a.counter + CTR_TODO, (uint32_t)__builtin_popcountll(todo));
    if (lane == 0u && done != 0ull) atomicAdd(a.counter + CTR_CLEAN, (uint32_t)__builtin_popcountll(done));
}

}  // namespace pzg
