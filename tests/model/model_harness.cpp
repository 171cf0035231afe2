// model_harness.cpp -- TEST INFRASTRUCTURE: compiles the kernel source (inflate_core.h) as a
// one-lane host program (PZG_WAVE == 1) so the CPU test-suite can fuzz the kernel's control
// logic, table construction and error ordering against the oracle without a GPU.
// Never linked into libpzg.so; the product has no CPU path.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../pure_zlib_amd/csrc/inflate_core.h"
#include "../../pure_zlib_amd/csrc/bundle_core.h"

struct pzm_result {
    int32_t status;
    uint32_t detail0, detail1, adler;
    uint64_t out_len, in_used;
};

// what crc32_verify_kernel computes (bit by bit here; the kernel's slicing and GF(2) fold are GPU-tested)
static uint32_t crc32_bits(const uint8_t *p, uint64_t n)
{
    uint32_t reg = 0xffffffffu;
    for (uint64_t i = 0; i < n; ++i) {
        reg ^= p[i];
        for (int k = 0; k < 8; ++k) reg = (reg >> 1) ^ (0xedb88320u & (0u - (reg & 1u)));
    }
    return ~reg;
}

// the scratch a model "wave" keeps from one stream to the next (its profile is in it: strip_profile_learn)
template <int RB, bool GZ>
static uint32_t *&kept_scratch()
{
    static uint32_t *kept = nullptr;
    return kept;
}

template <int RB, bool GZ = false>
static void run_one(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t cap, pzm_result *r)
{
    const bool gzip = GZ;
    auto *lds = (pzg::WaveLds<RB> *)aligned_alloc(16, sizeof(pzg::WaveLds<RB>));
    memset(lds, 0xA5, sizeof(*lds));  // LDS is not zero-initialised on the device either
    // pad the input on both sides: the bit reader loads whole aligned dwords
    uint8_t *buf = (uint8_t *)malloc(in_len + 16);
    memset(buf, 0xEE, in_len + 16);
    if (in_len) memcpy(buf + 8, in, in_len);
    pzg::Decoder<RB, GZ> dec(*lds);
    pzg::StreamResult sr;
    // the wave's token scratch (strips); PZM_NO_STRIPS=1 in the environment: the windows alone, as on a launch without scratch
    // (it outlives the call, as a persistent wave's scratch outlives its streams: the wave's profile -- strip_profile_learn() -- is in it;
    // PZM_FRESH_SCRATCH=1: a new one, filled with garbage, every call)
    uint32_t *&kept = kept_scratch<RB, GZ>();
    const bool fresh = getenv("PZM_FRESH_SCRATCH") != nullptr;
    uint32_t *strip = nullptr;
    if (!getenv("PZM_NO_STRIPS")) {
        if (fresh || !kept) {
            strip = (uint32_t *)malloc(sizeof(uint32_t) * pzg::Decoder<RB, GZ>::STRIP_WORDS);
            memset(strip, 0xC3, sizeof(uint32_t) * pzg::Decoder<RB, GZ>::STRIP_WORDS);
            if (!fresh) kept = strip;
        } else
            strip = kept;
    }
    dec.strip = strip;
    dec.run(buf + 8, in_len, out, cap, &sr);
    if (fresh) free(strip);
    r->status = sr.status;
    r->detail0 = sr.detail0;
    r->detail1 = sr.detail1;
    r->adler = sr.adler;
    r->out_len = sr.out_len;
    r->in_used = sr.in_used;
    if (gzip && (sr.status == pzg::ST_OK || sr.status == pzg::ST_GZIP_ISIZE)) {  // the verify pass of the gzip launch
        const uint32_t ours = crc32_bits(out, sr.out_len < cap ? sr.out_len : cap);
        r->adler = ours;
        if (sr.out_len > cap) {
            // (not stored: nothing to check; a length mismatch stays what it is)
        } else if (sr.gz_crc != ours) {
            r->status = pzg::ST_CHECKSUM;
            r->detail0 = sr.gz_crc;
            r->detail1 = ours;
        } else if (sr.status == pzg::ST_OK) {
            r->detail0 = r->detail1 = 0;
        }
    }
    free(buf);
    free(lds);
}

// the resumable instance (decompressIncremental): one call on a decoder whose slot (ResumeState + LDS image + for the small
// rings its 32 KiB history) the caller keeps in `state` (pzm_resume_state_bytes_rb(ring) bytes, zeroed for a fresh decoder)
template <int RB>
static int resume_feed(uint8_t *state, const uint8_t *in, uint64_t in_len, uint32_t final_input, uint8_t *out, uint64_t cap, pzm_result *r,
                       uint32_t *chunks)
{
    auto *lds = (pzg::WaveLds<RB> *)aligned_alloc(16, (sizeof(pzg::WaveLds<RB>) + 15u) & ~(size_t)15u);
    memset(lds, 0xA5, sizeof(*lds));
    uint8_t *buf = (uint8_t *)malloc(in_len + 16);
    memset(buf, 0xEE, in_len + 16);
    if (in_len) memcpy(buf + 8, in, in_len);
    pzg::Decoder<RB, false, true> dec(*lds);
    pzg::StreamResult sr;
    // the wave's scratch (strips), as in run_one(); PZM_NO_STRIPS=1: the windows alone
    uint32_t *strip = getenv("PZM_NO_STRIPS") ? nullptr : (uint32_t *)malloc(sizeof(uint32_t) * pzg::Decoder<RB, false, true>::STRIP_WORDS);
    if (strip) memset(strip, 0xC3, sizeof(uint32_t) * pzg::Decoder<RB, false, true>::STRIP_WORDS);
    dec.strip = strip;
    dec.run_resume((pzg::ResumeState *)state, (uint32_t *)(state + pzg::ResumeSlot<RB>::IMAGE_OFF), state + pzg::ResumeSlot<RB>::HIST_OFF, buf + 8,
                   in_len, out, cap, final_input, &sr, chunks);
    r->status = sr.status;
    r->detail0 = sr.detail0;
    r->detail1 = sr.detail1;
    r->adler = sr.adler;
    r->out_len = sr.out_len;
    r->in_used = sr.in_used;
    free(strip);
    free(buf);
    free(lds);
    return 0;
}

extern "C" {

int pzm_decompress(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t cap, int ring_bits, pzm_result *r)
{
    if (ring_bits == 15) run_one<15>(in, in_len, out, cap, r);
    else if (ring_bits == 14) run_one<14>(in, in_len, out, cap, r);
    else if (ring_bits == 13) run_one<13>(in, in_len, out, cap, r);
    else if (ring_bits == 12) run_one<12>(in, in_len, out, cap, r);
    else if (ring_bits == 11) run_one<11>(in, in_len, out, cap, r);
    else return -1;
    if (r->status == pzg::ST_RETRY_FULL_RING) run_one<15>(in, in_len, out, cap, r);  // what the fixup launch does
    return 0;
}

int pzm_decompress_gzip(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t cap, int ring_bits, pzm_result *r)
{
    if (ring_bits == 15) run_one<15, true>(in, in_len, out, cap, r);
    else if (ring_bits == 14) run_one<14, true>(in, in_len, out, cap, r);
    else if (ring_bits == 13) run_one<13, true>(in, in_len, out, cap, r);
    else if (ring_bits == 12) run_one<12, true>(in, in_len, out, cap, r);
    else if (ring_bits == 11) run_one<11, true>(in, in_len, out, cap, r);
    else return -1;
    if (r->status == pzg::ST_RETRY_FULL_RING) run_one<15, true>(in, in_len, out, cap, r);
    return 0;
}

uint32_t pzm_resume_state_bytes_rb(int rb)
{
    return (uint32_t)(rb == 15 ? pzg::ResumeSlot<15>::BYTES : rb == 12 ? pzg::ResumeSlot<12>::BYTES : pzg::ResumeSlot<11>::BYTES);
}
uint32_t pzm_resume_state_bytes(void) { return pzm_resume_state_bytes_rb(15); }

int pzm_resume_feed_rb(int rb, uint8_t *state, const uint8_t *in, uint64_t in_len, uint32_t final_input, uint8_t *out, uint64_t cap, pzm_result *r,
                       uint32_t *chunks)
{
    if (rb == 15) return resume_feed<15>(state, in, in_len, final_input, out, cap, r, chunks);
    if (rb == 12) return resume_feed<12>(state, in, in_len, final_input, out, cap, r, chunks);
    if (rb == 11) return resume_feed<11>(state, in, in_len, final_input, out, cap, r, chunks);
    return -1;
}
int pzm_resume_feed(uint8_t *state, const uint8_t *in, uint64_t in_len, uint32_t final_input, uint8_t *out, uint64_t cap, pzm_result *r,
                    uint32_t *chunks)
{
    return pzm_resume_feed_rb(15, state, in, in_len, final_input, out, cap, r, chunks);
}

// with a preset dictionary (PZG_FDICT extension): the 32 KiB-ring instance
int pzm_decompress_dict(const uint8_t *in, uint64_t in_len, const uint8_t *dict, uint32_t dict_len, uint8_t *out, uint64_t cap, pzm_result *r)
{
    auto *lds = (pzg::WaveLds<15> *)aligned_alloc(16, sizeof(pzg::WaveLds<15>));
    memset(lds, 0xA5, sizeof(*lds));
    uint8_t *buf = (uint8_t *)malloc(in_len + 16);
    memset(buf, 0xEE, in_len + 16);
    if (in_len) memcpy(buf + 8, in, in_len);
    pzg::Decoder<15, false> dec(*lds);
    pzg::StreamResult sr;
    dec.run(buf + 8, in_len, out, cap, &sr, dict, dict_len);
    r->status = sr.status;
    r->detail0 = sr.detail0;
    r->detail1 = sr.detail1;
    r->adler = sr.adler;
    r->out_len = sr.out_len;
    r->in_used = sr.in_used;
    free(buf);
    free(lds);
    return 0;
}

// The bundles (bundle_core.h): up to 64 streams, lane k decodes stream k; what a lane does not take comes back with status
// ST_BUNDLE_TODO (103: the ordinary kernel's).
int pzm_bundle(const uint8_t *const *ins, const uint64_t *in_lens, uint8_t *const *outs, const uint64_t *caps, uint32_t n, pzm_result *r)
{
    if (n > 64u) return -1;
    typedef pzg::Bundle B;
    auto *lds = (pzg::BundleLds *)aligned_alloc(16, sizeof(pzg::BundleLds));
    memset(lds, 0xA5, sizeof(*lds));
    uint8_t *bufs[64] = {};
    static uint32_t common[4] = {0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu};
    static uint8_t nowhere[16];
    B::In bi;
    B::Out bo;
    for (uint32_t k = 0; k < 64u; ++k) {
        const bool have = k < n;
        const uint64_t len = have ? in_lens[k] : 0, cap = have ? caps[k] : 0;
        if (have) {
            bufs[k] = (uint8_t *)malloc(len + 16 + (k & 3u));
            memset(bufs[k], 0xEE, len + 16 + (k & 3u));
            if (len) memcpy(bufs[k] + 8 + (k & 3u), ins[k], len);  // (every alignment of a stream's first byte)
        }
        const bool on = have && len >= 8u && len < B::MAX_BYTES && cap < B::MAX_BYTES;
        bi.IN.v[k] = have ? bufs[k] + 8 + (k & 3u) : (const uint8_t *)common;
        bi.OUT.v[k] = have ? outs[k] : nowhere;
        bi.LEN.v[k] = on ? (uint32_t)len : 0u;
        bi.CAP.v[k] = on ? (uint32_t)cap : 0u;
        bi.ON.v[k] = on ? 1u : 0u;
    }
    B::run(*lds, bi, common, bo);
    for (uint32_t k = 0; k < n; ++k) {
        memset(&r[k], 0, sizeof(r[k]));
        if (bo.STATE.v[k] != B::BS_CLEAN) {
            r[k].status = pzg::ST_BUNDLE_TODO;
            continue;
        }
        r[k].status = (int32_t)bo.STATUS.v[k];
        r[k].detail0 = bo.D0.v[k];
        r[k].detail1 = bo.D1.v[k];
        r[k].adler = bo.ADLER.v[k];
        r[k].out_len = bo.OLEN.v[k];
        r[k].in_used = bo.USED.v[k];
    }
    for (uint32_t k = 0; k < n; ++k) free(bufs[k]);
    free(lds);
    return 0;
}

// Test hook: overwrite the profile words of the ring's kept scratch -- 80 dwords from PROF_OFF on: 64 quantiles,
// the span's extent, its token count, the magic word (pass pzm_profile_magic() for a profile the kernel will trust), the streams
// to skip, the back-off level -- so that the next stream is laid out by them.
uint32_t pzm_profile_magic(void) { return pzg::Decoder<11>::PROF_MAGIC; }
int pzm_poke_profile(int ring_bits, const uint32_t *words)
{
    if (ring_bits != 11 && ring_bits != 15) return -1;
    uint32_t *&kept = ring_bits == 11 ? kept_scratch<11, false>() : kept_scratch<15, false>();
    const size_t n = ring_bits == 11 ? pzg::Decoder<11>::STRIP_WORDS : pzg::Decoder<15>::STRIP_WORDS;
    const size_t off = ring_bits == 11 ? pzg::Decoder<11>::PROF_OFF : pzg::Decoder<15>::PROF_OFF;
    if (!kept) {
        kept = (uint32_t *)malloc(sizeof(uint32_t) * n);
        memset(kept, 0xC3, sizeof(uint32_t) * n);
    }
    memcpy(kept + off, words, 80 * sizeof(uint32_t));
    return 0;
}

uint32_t pzm_lds_bytes(int ring_bits)
{
    return ring_bits == 15 ? sizeof(pzg::WaveLds<15>) : ring_bits == 14 ? sizeof(pzg::WaveLds<14>)
         : ring_bits == 13 ? sizeof(pzg::WaveLds<13>) : ring_bits == 12 ? sizeof(pzg::WaveLds<12>) : sizeof(pzg::WaveLds<11>);
}
}
