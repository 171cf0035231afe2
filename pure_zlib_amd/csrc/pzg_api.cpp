// pzg_api.cpp -- the C ABI of include/pzg.h: a thin HIP launcher around the kernels.
//
// No CPU decode path exists in this library.  Every compute entry point launches the gfx950
// kernels of pzg_kernels.hip; when HIP has no usable device the calls fail with PZG_RC_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
#include <new>
#include <string>

#include "../../include/pzg.h"
#include "pzg_launch.h"

namespace {

struct Arena {
    void *p = nullptr;
    size_t cap = 0;
};

constexpr uint32_t ADLER_MAX_WAVES = 8192;  // 256 CUs x 32 waves

}  // namespace

struct pzg_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::mutex mu;
    Arena a_in, a_out, a_meta, a_adler, a_gz;
    void *h_stage = nullptr;  // pinned host staging for the host-pointer path
    size_t h_stage_cap = 0;
    std::string last_error;
    void *prof_buf = nullptr;  // diagnostic builds only
    void *d_counter = nullptr; // stream-index counter of the persistent inflate waves
    hipStream_t s_up = nullptr, s_dn = nullptr;  // host-pointer path of big batches: H2D and D2H beside the kernel stream
    hipEvent_t ev_up[8] = {}, ev_k[8] = {}, ev_dn[8] = {};
    int ring_bits = PZG_DEFAULT_RING_BITS;
    int num_cus = 256;
};

namespace {

int hip_fail(pzg_ctx *ctx, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (ctx) ctx->last_error = buf;
    return PZG_RC_HIP_ERROR;
}

#define HIP_TRY(ctx, call)                                  \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int arena_reserve(pzg_ctx *ctx, Arena &a, size_t bytes)
{
    bytes = (bytes + 255u) & ~(size_t)255u;
    if (bytes == 0) bytes = 256;
    if (a.cap >= bytes) return PZG_RC_OK;
    if (a.p) HIP_TRY(ctx, hipFree(a.p));
    a.p = nullptr;
    a.cap = 0;
    hipError_t e = hipMalloc(&a.p, bytes);
    if (e != hipSuccess) {
        hip_fail(ctx, e, "hipMalloc");
        return PZG_RC_NO_MEMORY;
    }
    a.cap = bytes;
    return PZG_RC_OK;
}

// the two copy streams and the per-range events of the pipelined host-pointer path, created on first use
int ensure_pipeline(pzg_ctx *ctx)
{
    if (ctx->s_up) return PZG_RC_OK;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_dn, hipStreamNonBlocking));
    for (int c = 0; c < 8; ++c) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_up[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_k[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_dn[c], hipEventDisableTiming));
    }
    return PZG_RC_OK;
}

int launch_timed(pzg_ctx *ctx, const pzg::InflateArgs &args_in)
{
    pzg::InflateArgs args = args_in;
    args.counter = (uint32_t *)ctx->d_counter;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(ctx, pzg::launch_inflate(args, ctx->ring_bits, ctx->num_cus, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    ctx->timed = true;
    return PZG_RC_OK;
}

}  // namespace

extern "C" {

#if defined(PZG_PROFILE)
// diagnostic builds only: device buffer of 12 uint64 per stream the kernel fills with cycle counters
int pzg_prof_buffer(pzg_ctx *ctx, uint32_t n, uint64_t *host_out)
{
    if (!ctx->prof_buf) { if (hipMalloc(&ctx->prof_buf, 128u * 1048576u) != hipSuccess) return -1; }
    if (host_out) return hipMemcpy(host_out, ctx->prof_buf, 128u * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    return 0;
}
#endif

int pzg_init(int device, pzg_ctx **out)
{
    if (!out) return PZG_RC_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return PZG_RC_NO_DEVICE;
    if (device < 0 || device >= ndev) return PZG_RC_BAD_ARG;
    pzg_ctx *ctx = new (std::nothrow) pzg_ctx();
    if (!ctx) return PZG_RC_NO_MEMORY;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return PZG_RC_NO_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->num_cus = prop.multiProcessorCount;
        if (const char *e = getenv("PZG_RING_BITS")) {  // environment override of the default size class
            const int rb = atoi(e);
            if (rb >= 11 && rb <= 15) ctx->ring_bits = rb;
        }
    }
    if (hipMalloc(&ctx->d_counter, 256) != hipSuccess) {
        delete ctx;
        return PZG_RC_NO_MEMORY;
    }
    *out = ctx;
    return PZG_RC_OK;
}

void pzg_shutdown(pzg_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (Arena *a : {&ctx->a_in, &ctx->a_out, &ctx->a_meta, &ctx->a_adler, &ctx->a_gz})
        if (a->p) (void)hipFree(a->p);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (int c = 0; c < 8; ++c) {
        if (ctx->ev_up[c]) (void)hipEventDestroy(ctx->ev_up[c]);
        if (ctx->ev_k[c]) (void)hipEventDestroy(ctx->ev_k[c]);
        if (ctx->ev_dn[c]) (void)hipEventDestroy(ctx->ev_dn[c]);
    }
    if (ctx->s_up) (void)hipStreamDestroy(ctx->s_up);
    if (ctx->s_dn) (void)hipStreamDestroy(ctx->s_dn);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pzg_set_stream(pzg_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stream = (hipStream_t)hip_stream;  // NULL = the default (null) stream
    return PZG_RC_OK;
}

int pzg_reset_stream(pzg_ctx *ctx)
{
    if (!ctx) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->stream = ctx->own_stream;
    return PZG_RC_OK;
}

int pzg_set_option(pzg_ctx *ctx, int option, int64_t value)
{
    if (!ctx) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (option == PZG_OPT_RING_BITS && value >= 11 && value <= 15) {
        ctx->ring_bits = (int)value;
        return PZG_RC_OK;
    }
    return PZG_RC_BAD_ARG;
}

int pzg_sync(pzg_ctx *ctx)
{
    if (!ctx) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PZG_RC_OK;
}

int pzg_decompress_many(pzg_ctx *ctx, const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                        uint8_t *out_base, const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                        int32_t *status, uint32_t *detail, uint64_t *in_used, uint32_t *adler, uint32_t n,
                        uint32_t flags)
{
    if (!ctx) return PZG_RC_BAD_ARG;
    if (n == 0) return PZG_RC_OK;
    if (!in_base || !in_off || !in_len || !out_off || !out_cap || !out_len || !status) return PZG_RC_BAD_ARG;
    if ((flags & PZG_ASYNC) && !(flags & PZG_DEVICE_PTRS)) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    if (flags & PZG_DEVICE_PTRS) {
        if (!out_base) return PZG_RC_BAD_ARG;
        pzg::InflateArgs a{in_base, in_off, in_len, out_base, out_off, out_cap, out_len,
                           status,  detail, in_used, adler,   nullptr, nullptr, nullptr, n, 0, nullptr};
        if (flags & PZG_GZIP) {
            int rcg = arena_reserve(ctx, ctx->a_gz, 8 * (size_t)n);
            if (rcg != PZG_RC_OK) return rcg;
            a.gzip = 1;
            a.gz_expect = (uint32_t *)ctx->a_gz.p;
        }
#if defined(PZG_PROFILE)
        a.prof_out = (uint64_t *)ctx->prof_buf;
#endif
        int rc = launch_timed(ctx, a);
        if (rc != PZG_RC_OK) return rc;
        if (!(flags & PZG_ASYNC)) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return PZG_RC_OK;
    }

    // host-pointer path: stage the covering byte ranges through the context's arenas.  A big batch is cut into
    // index ranges that flow through three HIP streams -- H2D of range c+1, the kernel on range c, D2H of range
    // c-1 into pinned staging -- while host threads copy range c-2 out to the caller's extents.
    uint64_t in_lo = ~0ull, in_hi = 0, out_lo = ~0ull, out_hi = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (in_off[i] < in_lo) in_lo = in_off[i];
        if (in_off[i] + in_len[i] > in_hi) in_hi = in_off[i] + in_len[i];
        if (out_off[i] < out_lo) out_lo = out_off[i];
        if (out_off[i] + out_cap[i] > out_hi) out_hi = out_off[i] + out_cap[i];
    }
    const uint64_t in_bytes = in_hi - in_lo, out_bytes = out_hi - out_lo;
    if (out_bytes && !out_base) return PZG_RC_BAD_ARG;
    // keep every stream's address alignment (mod 16) on the device what it is on the host
    const uint32_t in_skew = (uint32_t)(((uintptr_t)in_base + in_lo) & 15u);
    const uint32_t out_skew = (uint32_t)(((uintptr_t)out_base + out_lo) & 15u);
    int rc;
    if ((rc = arena_reserve(ctx, ctx->a_in, in_bytes + 64)) != PZG_RC_OK) return rc;
    if ((rc = arena_reserve(ctx, ctx->a_out, out_bytes + 64)) != PZG_RC_OK) return rc;
    // meta layout: in_off | in_len | out_off | out_cap | out_len | in_used  (u64[n] each)
    //              | status | adler (32-bit [n] each) | detail u32[2n]
    const size_t N = n;
    const size_t m_in_off = 0, m_in_len = 8 * N, m_out_off = 16 * N, m_out_cap = 24 * N, m_out_len = 32 * N,
                 m_in_used = 40 * N, m_status = 48 * N, m_adler = 52 * N, m_detail = 56 * N, m_total = 64 * N;
    if ((rc = arena_reserve(ctx, ctx->a_meta, m_total)) != PZG_RC_OK) return rc;
    if ((flags & PZG_GZIP) && (rc = arena_reserve(ctx, ctx->a_gz, 8 * N)) != PZG_RC_OK) return rc;
    if (out_bytes && ctx->h_stage_cap < out_bytes) {
        if (ctx->h_stage) HIP_TRY(ctx, hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr;
        ctx->h_stage_cap = 0;
        hipError_t e = hipHostMalloc(&ctx->h_stage, out_bytes + 64, hipHostMallocDefault);
        if (e != hipSuccess) {
            hip_fail(ctx, e, "hipHostMalloc");
            return PZG_RC_NO_MEMORY;
        }
        ctx->h_stage_cap = out_bytes;
    }
    uint8_t *d_in = (uint8_t *)ctx->a_in.p + in_skew;
    uint8_t *d_out = (uint8_t *)ctx->a_out.p + out_skew;
    uint8_t *d_meta = (uint8_t *)ctx->a_meta.p;
    uint8_t *stage = (uint8_t *)ctx->h_stage;
    const bool trace = getenv("PZG_TRACE_HOST") != nullptr;
    const auto t_call0 = std::chrono::steady_clock::now();

    // index ranges: one for a small batch, four when there is enough to overlap (more only adds kernel tails:
    // measured 91 / 83 / 79 / 77 / 78 ms for 1 / 2 / 3 / 4 / 8 ranges on the 65,536 x 32 KiB batch)
    uint32_t nchunk = 1;
    if (n >= 4096u && in_bytes + out_bytes >= (256ull << 20)) nchunk = 4u;
    if (const char *e = getenv("PZG_HOST_RANGES")) {  // experiment knob
        const int v = atoi(e);
        if (v >= 1 && v <= 8 && (uint32_t)v <= n) nchunk = (uint32_t)v;
    }
    if (nchunk > 1 && ensure_pipeline(ctx) != PZG_RC_OK) nchunk = 1;
    hipStream_t s_k = ctx->stream;                               // kernels: the context's (or the caller's) stream
    hipStream_t s_up = nchunk > 1 ? ctx->s_up : ctx->stream;     // H2D
    hipStream_t s_dn = nchunk > 1 ? ctx->s_dn : ctx->stream;     // D2H
    uint32_t cthreads = 1;
    if (out_bytes >= (64ull << 20) && n >= 64u) {
        cthreads = std::thread::hardware_concurrency();
        cthreads = cthreads > 8u ? 8u : cthreads < 1u ? 1u : cthreads;
    }
    auto copy_out = [&](uint32_t lo, uint32_t hi) {  // per-stream copies out of staging: only [out_off, out_off + min(out_len, out_cap))
        auto part = [&](uint32_t a0, uint32_t a1) {
            for (uint32_t i = a0; i < a1; ++i) {
                const uint64_t nb = out_len[i] < out_cap[i] ? out_len[i] : out_cap[i];
                if (nb) memcpy(out_base + out_off[i], stage + (out_off[i] - out_lo), nb);
            }
        };
        const uint32_t m = hi - lo;
        if (cthreads == 1 || m < 64u) {
            part(lo, hi);
            return;
        }
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < cthreads; ++t)
            pool.emplace_back(part, lo + (uint32_t)((uint64_t)m * t / cthreads), lo + (uint32_t)((uint64_t)m * (t + 1) / cthreads));
        for (auto &th : pool) th.join();
    };

    HIP_TRY(ctx, hipMemcpyAsync(d_meta + m_in_off, in_off, 8 * N, hipMemcpyHostToDevice, s_up));
    HIP_TRY(ctx, hipMemcpyAsync(d_meta + m_in_len, in_len, 8 * N, hipMemcpyHostToDevice, s_up));
    HIP_TRY(ctx, hipMemcpyAsync(d_meta + m_out_off, out_off, 8 * N, hipMemcpyHostToDevice, s_up));
    HIP_TRY(ctx, hipMemcpyAsync(d_meta + m_out_cap, out_cap, 8 * N, hipMemcpyHostToDevice, s_up));
    struct Range {
        uint32_t lo, hi;
        uint64_t ilo, ihi, olo, ohi;
    };
    std::vector<Range> rg(nchunk);
    for (uint32_t c = 0; c < nchunk; ++c) {
        Range &r = rg[c];
        r.lo = (uint32_t)((uint64_t)n * c / nchunk);
        r.hi = (uint32_t)((uint64_t)n * (c + 1) / nchunk);
        r.ilo = r.olo = ~0ull;
        r.ihi = r.ohi = 0;
        for (uint32_t i = r.lo; i < r.hi; ++i) {
            if (in_off[i] < r.ilo) r.ilo = in_off[i];
            if (in_off[i] + in_len[i] > r.ihi) r.ihi = in_off[i] + in_len[i];
            if (out_off[i] < r.olo) r.olo = out_off[i];
            if (out_off[i] + out_cap[i] > r.ohi) r.ohi = out_off[i] + out_cap[i];
        }
    }
    for (uint32_t c = 0; c < nchunk; ++c) {
        const Range &r = rg[c];
        const size_t lo = r.lo, m = r.hi - r.lo;
        if (r.ihi > r.ilo)
            HIP_TRY(ctx, hipMemcpyAsync(d_in + (r.ilo - in_lo), in_base + r.ilo, r.ihi - r.ilo, hipMemcpyHostToDevice, s_up));
        if (nchunk > 1) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_up[c], s_up));
            HIP_TRY(ctx, hipStreamWaitEvent(s_k, ctx->ev_up[c], 0));
        }
        pzg::InflateArgs a{};
        a.in_base = d_in - in_lo;  // offsets stay the caller's
        a.out_base = d_out - out_lo;
        a.in_off = (const uint64_t *)(d_meta + m_in_off) + lo;
        a.in_len = (const uint64_t *)(d_meta + m_in_len) + lo;
        a.out_off = (const uint64_t *)(d_meta + m_out_off) + lo;
        a.out_cap = (const uint64_t *)(d_meta + m_out_cap) + lo;
        a.out_len = (uint64_t *)(d_meta + m_out_len) + lo;
        a.in_used = (uint64_t *)(d_meta + m_in_used) + lo;
        a.status = (int32_t *)(d_meta + m_status) + lo;
        a.adler = (uint32_t *)(d_meta + m_adler) + lo;
        a.detail = (uint32_t *)(d_meta + m_detail) + 2 * lo;
        a.order = nullptr;
        a.prof_out = nullptr;
#if defined(PZG_PROFILE)
        a.prof_out = ctx->prof_buf ? (uint64_t *)ctx->prof_buf + 16 * lo : nullptr;
#endif
        a.n = (uint32_t)m;
        if (flags & PZG_GZIP) {
            a.gzip = 1;
            a.gz_expect = (uint32_t *)ctx->a_gz.p + 2 * lo;
        }
        a.counter = (uint32_t *)ctx->d_counter;
        if (c == 0) HIP_TRY(ctx, hipEventRecord(ctx->ev0, s_k));
        HIP_TRY(ctx, pzg::launch_inflate(a, ctx->ring_bits, ctx->num_cus, s_k));
        if (c + 1 == nchunk) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev1, s_k));
            ctx->timed = true;
        }
        if (nchunk > 1) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k[c], s_k));
            HIP_TRY(ctx, hipStreamWaitEvent(s_dn, ctx->ev_k[c], 0));
        }
        HIP_TRY(ctx, hipMemcpyAsync(out_len + lo, a.out_len, 8 * m, hipMemcpyDeviceToHost, s_dn));
        HIP_TRY(ctx, hipMemcpyAsync(status + lo, a.status, 4 * m, hipMemcpyDeviceToHost, s_dn));
        if (in_used) HIP_TRY(ctx, hipMemcpyAsync(in_used + lo, a.in_used, 8 * m, hipMemcpyDeviceToHost, s_dn));
        if (adler) HIP_TRY(ctx, hipMemcpyAsync(adler + lo, a.adler, 4 * m, hipMemcpyDeviceToHost, s_dn));
        if (detail) HIP_TRY(ctx, hipMemcpyAsync(detail + 2 * lo, a.detail, 8 * m, hipMemcpyDeviceToHost, s_dn));
        // one D2H transfer of the range's covering bytes into pinned staging
        if (r.ohi > r.olo)
            HIP_TRY(ctx, hipMemcpyAsync(stage + (r.olo - out_lo), d_out + (r.olo - out_lo), r.ohi - r.olo, hipMemcpyDeviceToHost, s_dn));
        if (nchunk > 1) HIP_TRY(ctx, hipEventRecord(ctx->ev_dn[c], s_dn));
        // while that is in flight: hand the previous range to the caller
        if (nchunk > 1 && c >= 1) {
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev_dn[c - 1]));
            if (out_bytes) copy_out(rg[c - 1].lo, rg[c - 1].hi);
        }
    }
    if (nchunk > 1) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_dn[nchunk - 1]));
    } else {
        HIP_TRY(ctx, hipStreamSynchronize(s_dn));
    }
    if (out_bytes) copy_out(rg[nchunk - 1].lo, rg[nchunk - 1].hi);
    HIP_TRY(ctx, hipStreamSynchronize(s_k));
    if (trace)
        fprintf(stderr, "[pzg] host path: %u streams, %.1f MiB in, %.1f MiB out, %u range(s), %u copy thread(s): %.1f ms\n", n,
                in_bytes / 1048576.0, out_bytes / 1048576.0, nchunk, cthreads,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call0).count());
    return PZG_RC_OK;
}

int pzg_decompress(pzg_ctx *ctx, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                   uint64_t *out_len, int32_t *status, uint32_t detail[2], uint64_t *in_used)
{
    if (!out_len || !status) return PZG_RC_BAD_ARG;
    static const uint8_t dummy_in = 0;
    static uint8_t dummy_out = 0;
    const uint64_t zero = 0;
    return pzg_decompress_many(ctx, in ? in : &dummy_in, &zero, &in_len, out ? out : &dummy_out, &zero, &out_cap,
                               out_len, status, detail, in_used, nullptr, 1, 0);
}

int pzg_adler32(pzg_ctx *ctx, const uint8_t *buf, uint64_t len, uint32_t init, uint32_t *out, uint32_t flags)
{
    if (!ctx || !out || (!buf && len)) return PZG_RC_BAD_ARG;
    if ((flags & PZG_ASYNC) && !(flags & PZG_DEVICE_PTRS)) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = arena_reserve(ctx, ctx->a_adler, 12 * (size_t)(ADLER_MAX_WAVES + 4) + 16)) != PZG_RC_OK) return rc;
    uint32_t *partials = (uint32_t *)ctx->a_adler.p;
    uint32_t *d_res = partials + 3 * (size_t)(ADLER_MAX_WAVES + 4);
    hipStream_t s = ctx->stream;
    if (flags & PZG_DEVICE_PTRS) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, s));
        HIP_TRY(ctx, pzg::launch_adler32(buf, len, init, partials, ADLER_MAX_WAVES, out, s));
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, s));
        ctx->timed = true;
        if (!(flags & PZG_ASYNC)) HIP_TRY(ctx, hipStreamSynchronize(s));
        return PZG_RC_OK;
    }
    const uint32_t skew = (uint32_t)((uintptr_t)buf & 15u);
    if ((rc = arena_reserve(ctx, ctx->a_in, len + 64)) != PZG_RC_OK) return rc;
    uint8_t *d_buf = (uint8_t *)ctx->a_in.p + skew;
    if (len) HIP_TRY(ctx, hipMemcpyAsync(d_buf, buf, len, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, s));
    HIP_TRY(ctx, pzg::launch_adler32(d_buf, len, init, partials, ADLER_MAX_WAVES, d_res, s));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, s));
    ctx->timed = true;
    HIP_TRY(ctx, hipMemcpyAsync(out, d_res, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return PZG_RC_OK;
}

double pzg_last_kernel_ms(pzg_ctx *ctx)
{
    if (!ctx) return -1.0;
    std::lock_guard<std::mutex> g(ctx->mu);
    if (!ctx->timed) return -1.0;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

const char *pzg_strerror(int rc)
{
    switch (rc) {
    case PZG_RC_OK: return "ok";
    case PZG_RC_BAD_ARG: return "bad argument";
    case PZG_RC_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case PZG_RC_HIP_ERROR: return "HIP runtime error";
    case PZG_RC_NO_MEMORY: return "out of memory";
    default: return "unknown return code";
    }
}

const char *pzg_last_error(pzg_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

uint32_t pzg_version(void) { return (PZG_VERSION_MAJOR << 16) | PZG_VERSION_MINOR; }

}  // extern "C"
