// pzg_launch.h -- kernel argument block and launcher prototypes shared by pzg_kernels.hip and pzg_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pzg {

struct InflateArgs {
    const uint8_t *in_base;
    const uint64_t *in_off;   // n
    const uint64_t *in_len;   // n
    uint8_t *out_base;
    const uint64_t *out_off;  // n
    const uint64_t *out_cap;  // n
    uint64_t *out_len;        // n
    int32_t *status;          // n
    uint32_t *detail;         // 2n or null
    uint64_t *in_used;        // n or null
    uint32_t *adler;          // n or null
    const uint32_t *order;    // optional launch permutation (n) or null
    uint32_t *counter;        // device word the persistent waves draw stream indices from (zeroed per launch)
    uint64_t *prof_out;       // diagnostic builds only (-DPZG_PROFILE): 12 counters per stream, else null
    uint32_t n;
    uint32_t gzip;            // 0: zlib streams (RFC 1950, the reference's format); 1: gzip members (RFC 1952, an extension)
    uint32_t *gz_expect;      // gzip only: 2n words of device scratch: the CRC-32 the output must have (from the member trailers)
    // extension (PZG_FDICT): preset dictionaries, dict_base + dict_off[i] .. + dict_len[i] (0: none for stream i); null: none at all
    const uint8_t *dict_base;
    const uint64_t *dict_off;
    const uint64_t *dict_len;
    // token scratch of the launch's stream-waves (inflate_core.h strip_span): inflate_strip_bytes() bytes, or null -- the
    // kernels then decode by windows alone.  Must not be shared with a launch that may run at the same time.
    uint32_t *strip;
    uint32_t strip_waves;     // stream-waves (workgroups 0 .. strip_waves - 1) that own a slice of it; the others decode by windows alone
    // bundles (round 6; pzg_bundle_kernel.h): nonzero -- the launch's small streams of the fixed code are decoded 64 to a wave, one
    // lane per stream, before the ordinary kernel takes what is left.  zlib streams without dictionaries only.
    uint32_t bundle;
    uint32_t *bundle_report;  // null, or a word of page-locked HOST memory: the launch's last kernel leaves 1 + the streams the bundles decoded there
};

// one batched call of the resumable decoder (decompressIncremental): decoder i continues from its ResumeState
struct ResumeArgs {
    uint8_t *state_base;      // n slots of state_stride bytes: ResumeState + the LDS image (zeroed = a fresh decoder)
    uint64_t state_stride;
    const uint8_t *in_base;
    const uint64_t *in_off;   // n: the unconsumed tail of the last call followed by the new input
    const uint64_t *in_len;   // n
    const uint8_t *final_in;  // n or null: nonzero = no more input will follow (running out is an error then)
    uint8_t *out_base;
    const uint64_t *out_off;  // n: this call's output room
    const uint64_t *out_cap;  // n (>= 4096)
    uint64_t *out_len;        // n: bytes delivered by this call
    int32_t *status;          // n: PZG_DEC_NEED_INPUT / PZG_DEC_OUT_FULL / PZG_OK (done) / PZG_E_*
    uint32_t *detail;         // 2n or null
    uint64_t *in_used;        // n: input bytes the decoder is done with
    uint32_t *adler;          // n or null
    uint32_t *chunks;         // n: 32 KiB chunks the reference would have published so far (cumulative)
    uint32_t *counter;
    uint32_t n;
    // optional (null: off): every decoder also packs what it delivered behind the others' -- dense + dense_region + 16 * (the value
    // of *dense_cursor it drew), reported in dense_off[i] -- so that the host fetches one linear span instead of mostly empty rooms
    uint8_t *dense;
    uint64_t dense_region;
    uint32_t *dense_cursor;  // zeroed by the launcher's caller; counts 16-byte units
    uint64_t *dense_off;     // n
    // the stream-waves' scratch (inflate_core.h strip_span), resume_strip_wave_bytes() per wave, or null: workgroups 0 .. strip_waves - 1
    // own a slice, the others -- and all of them without it -- decode by windows alone.  Not shared with a launch that may run at the same time.
    uint32_t *strip;
    uint32_t strip_waves;
};
hipError_t launch_resume(const ResumeArgs &a, int num_cus, hipStream_t stream);
size_t resume_state_bytes();   // one decoder's slot: ResumeState + LDS image
size_t resume_scalar_bytes();  // ... its ResumeState part (zeroing it makes the decoder fresh)
size_t resume_strip_wave_bytes();                      // one stream-wave's slice of ResumeArgs::strip
uint32_t resume_launch_waves(int num_cus, uint32_t n);  // workgroups of a launch over n decoders

hipError_t launch_inflate(const InflateArgs &a, int ring_bits, int num_cus, hipStream_t stream);
size_t inflate_strip_bytes(int ring_bits, int num_cus, uint32_t n, uint32_t gzip);  // what InflateArgs::strip holds when every stream-wave of that launch owns a slice
size_t inflate_strip_wave_bytes();  // ... one stream-wave's slice
// the profiles of `waves` stream-wave slices of a strips' scratch switched off (no stream consults or counts down its wave's profile
// from then on) or on again (inflate_core.h strip_profile_layout: the words behind the quantiles)
hipError_t launch_profile_switch(uint32_t *strip, uint32_t waves, bool off, hipStream_t stream);
// the gzip instances (pzg_kernels_b.hip): `waves` workgroups of the ring's kernel, or of the fixup pass
hipError_t launch_inflate_gzip(const InflateArgs &a, int ring_bits, bool fixup, uint32_t waves, hipStream_t stream);

// partials: 3 * 4 * ceil(max_waves / 4) uint32 of device scratch
hipError_t launch_adler32(const uint8_t *buf, uint64_t len, uint32_t init, uint32_t *partials, uint32_t max_waves,
                          uint32_t *out, hipStream_t stream);

// Adler-32 of n buffers base + off[i] .. + len[i] (device memory), one wave per buffer
hipError_t launch_adler32_many(const uint8_t *base, const uint64_t *off, const uint64_t *len, uint32_t *out, uint32_t n, int num_cus,
                               hipStream_t stream);

// order[0..n): stream indices, longest (by out_cap) first; scratch: 128 uint32 of device memory
hipError_t launch_order(const uint64_t *out_cap, uint32_t n, uint32_t *order, uint32_t *scratch, hipStream_t stream);

}  // namespace pzg
