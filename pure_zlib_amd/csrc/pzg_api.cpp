// pzg_api.cpp -- the C ABI of include/pzg.h: a thin HIP launcher around the kernels.
//
// No CPU decode path exists in this library.  Every compute entry point launches the gfx950
// kernels of pzg_kernels.hip; when HIP has no usable device the calls fail with PZG_RC_NO_DEVICE.
//
// Structure:
//   pzg_ctx      one or more device shards (pzg_init: one device; pzg_init_mask: every device of the mask)
//   Shard        one device: the launch stream of the device-pointer path + a few Lanes
//   Lane         one independent host-buffer pipeline: three HIP streams (H2D, kernels, D2H), double-buffered device
//                arenas and pinned staging slots.  A host-pointer call takes a free lane of its shard, so calls from
//                several threads overlap instead of queueing behind one mutex.
// Host-pointer batches are packed (no gaps travel over PCIe), launched longest stream first, and flow through the
// lane in index ranges: pack + H2D of range c+1, the kernel on range c and D2H + copy-out of range c-1 overlap, and
// the two PCIe directions run on their own streams.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pzg.h"
#include "pzg_helpers.h"
#include "pzg_launch.h"

namespace {

// A helper thread of one call that is joined whatever way the call ends (a std::thread destroyed while joinable ends
// the process): `wake` makes the thread's loop end -- it sets the call's stop flags under the call's mutex and notifies -- and
// runs first when the guard dies while the thread is still running (an early return, an exception on the issuing thread).
// start() returns false when the system has no thread to give: the caller then runs the stage inline.
struct CallThread {
    std::thread t;
    std::function<void()> wake;
    template <class F>
    bool start(F &&body)
    {
#if defined(PZG_LAB) && defined(PZG_LAB_NO_CALL_THREADS)  // (lab builds only, tests/test_gpu_api.py: as if the system had no thread to give)
        (void)body;
        return false;
#endif
        try {
            t = std::thread(std::forward<F>(body));
            return true;
        } catch (const std::system_error &) {
            return false;
        }
    }
    void join()
    {
        if (t.joinable()) t.join();
    }
    ~CallThread()
    {
        if (t.joinable()) {
            if (wake) wake();
            t.join();
        }
    }
};

struct Arena {
    void *p = nullptr;
    size_t cap = 0;
    bool prof_off = false;  // (a strips' scratch) its waves' profiles are switched off: PZG_OPT_PROFILE
};
struct Pinned {
    uint8_t *p = nullptr;
    size_t cap = 0;
    bool pageable = false;  // hipHostMalloc refused (locked-memory limit): plain malloc, the copies then go the runtime's slower way
};

constexpr uint32_t ADLER_MAX_WAVES = 8192;  // 256 CUs x 32 waves
constexpr int LANES_PER_SHARD = 4;
constexpr int STRIP_SLOTS = 2;
constexpr int COUNTER_SLOTS = 16;           // device-pointer launches in flight on one shard, each with its own counter
constexpr size_t RANGE_MAX_OUT = 256ull << 20;  // a range's packed output stays below this (bounds the pinned staging)

constexpr int NSLOT = 3;  // ranges in flight per lane: one being packed / uploaded, one decoding, one downloading / copied out
struct Lane {
    hipStream_t s_k = nullptr, s_up = nullptr, s_dn = nullptr;
    hipEvent_t ev_up[NSLOT] = {}, ev_k[NSLOT] = {}, ev_dn[NSLOT] = {};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // first / last kernel of the call this lane is running (pzg_last_kernel_ms)
    Arena d_in[NSLOT], d_out[NSLOT], d_meta[NSLOT], d_gz[NSLOT], d_ord[NSLOT];
    Arena d_strip;  // the kernels' token scratch: the lane's launches all go to s_k, one after the other
    Pinned h_in[NSLOT], h_out[NSLOT], h_meta[NSLOT];
    uint32_t *d_counter = nullptr;
    std::mutex mu;
    bool ready = false;
};

struct Shard {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the device-pointer path launches here (pzg_set_stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the last device-pointer launch (recorded under `mu`, so always a pair)
    bool timed = false;
    double host_ms = -1.0;           // kernel span of the last host-pointer call that finished (measured by that call itself)
    bool last_was_host = false;
    // Per-launch resources of the device-pointer path: launches on different user streams (pzg_set_stream in between,
    // PZG_ASYNC) may overlap, so each takes the next of COUNTER_SLOTS slots -- its own work counter, launch permutation
    // and expected-CRC array.  A slot comes round again after COUNTER_SLOTS launches.
    uint32_t *d_counters = nullptr;  // COUNTER_SLOTS x 64 words
    std::atomic<uint32_t> next_counter{0};
    Arena a_adler, a_scratch, a_dec;
    Arena a_gz[COUNTER_SLOTS], a_order[COUNTER_SLOTS];
    // ... and the kernels' token scratch (hundreds of MiB for a launch that fills the chip): STRIP_SLOTS of them, a launch
    // that takes one in use waits (on its own stream) for the launch that used it last
    // bundles (pzg_bundle_kernel.h): what the last probing launch found, fetched behind it -- a launch that decoded no stream by
    // bundles (a batch of dynamic-code streams) lets the next 2, 6, 14 ... 64 launches go without the probe
    uint32_t *h_bundle = nullptr;  // page-locked: [0] streams the last probing launch decoded by bundles
    hipEvent_t ev_bundle = nullptr;
    bool bundle_pending = false;
    uint32_t bundle_skip = 0, bundle_backoff = 0;
    Arena a_strip[STRIP_SLOTS];
    hipEvent_t ev_strip[STRIP_SLOTS] = {};
    bool strip_used[STRIP_SLOTS] = {};
    uint32_t next_strip = 0;
    Lane lanes[LANES_PER_SHARD];
    std::atomic<uint32_t> next_lane{0};
    std::mutex mu;  // device-pointer path bookkeeping (stream pointer, arenas above, timing events)
};

}  // namespace

struct pzg_ctx {
    // Lifetime (include/pzg.h "Lifetimes"): the caller's handle is one reference, every live pzg_decoder another; the
    // context is freed when the last of them goes.  `closed` is set by pzg_shutdown: calls on the handle after that fail.
    std::atomic<int> refs{1};
    std::atomic<bool> closed{false};
    std::vector<std::unique_ptr<Shard>> shards;
    int ring_bits = PZG_DEFAULT_RING_BITS;
    std::mutex err_mu;
    std::string last_error;
    void *prof_buf = nullptr;  // diagnostic builds only
    std::unique_ptr<Helpers> helpers;
    std::atomic<uint64_t> scratch_cap{0};  // PZG_OPT_SCRATCH_BYTES: the kernels' scratch per device, at most (0: no limit)
    std::atomic<int> bundles{1};           // PZG_OPT_BUNDLES
    std::atomic<int> profile{1};           // PZG_OPT_PROFILE
};

namespace {

int hip_fail(pzg_ctx *ctx, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (ctx) {
        std::lock_guard<std::mutex> g(ctx->err_mu);
        ctx->last_error = buf;
    }
    return PZG_RC_HIP_ERROR;
}

#define HIP_TRY(ctx, call)                                     \
    do {                                                       \
        hipError_t e_ = (call);                                \
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

// The kernels' scratch for one launch of n streams into arena `a`: as many whole stream-wave slices as the launch can use and the
// context's cap (PZG_OPT_SCRATCH_BYTES, shared evenly by the arenas a shard can hold) allows.  Failing is tolerated -- the
// launch then decodes by windows alone, slower and never wrong -- so nothing is recorded as an error.
void strip_for_launch(pzg_ctx *ctx, Arena &a, int num_cus, uint32_t n, uint32_t gzip, pzg::InflateArgs &args);

void strip_for_launch(pzg_ctx *ctx, Arena &a, int num_cus, uint32_t n, uint32_t gzip, pzg::InflateArgs &args)
{
    args.strip = nullptr;
    args.strip_waves = 0;
    const size_t per_wave = pzg::inflate_strip_wave_bytes();
    size_t want = pzg::inflate_strip_bytes(ctx->ring_bits, num_cus, n, gzip);
    const uint64_t cap = ctx->scratch_cap.load();
    if (cap != 0) {  // a shard holds STRIP_SLOTS arenas for device-pointer launches and one per host-path lane: an even share each
        const size_t share = (size_t)(cap / (uint64_t)(STRIP_SLOTS + LANES_PER_SHARD)) / per_wave * per_wave;
        if (want > share) want = share;
        if (a.cap > share + 255u) {  // (held from before the cap was set: given back)
            (void)hipFree(a.p);
            a.p = nullptr;
            a.cap = 0;
        }
    }
    if (want < per_wave) return;
    if (a.cap < want) {
        if (a.p) (void)hipFree(a.p);
        a.p = nullptr;
        a.cap = 0;
        void *q = nullptr;
        if (hipMalloc(&q, (want + 255u) & ~(size_t)255u) != hipSuccess) {
            (void)hipGetLastError();  // (tolerated: the launch decodes by windows alone)
            return;
        }
        a.p = q;
        a.cap = (want + 255u) & ~(size_t)255u;
        a.prof_off = false;  // (as the allocator left it: no profile yet)
    }
    args.strip = (uint32_t *)a.p;
    args.strip_waves = (uint32_t)(want / per_wave);
}

// Bundles (pzg_bundle_kernel.h) pay when the batch has more bundles than the chip has SIMDs to run them on: a lane decodes its
// stream token by token, so a bundle takes as long as its longest stream whether 64 lanes are busy or one -- measured on 4 KiB
// streams: ~0.9 ms a bundle against ~0.16 ms a stream for a stream-wave of the ordinary kernel, 6,656 of which run side by side.
constexpr uint32_t BUNDLE_MIN_STREAMS = 32768u;

void pinned_release(Pinned &a)
{
    if (a.p) {
        if (a.pageable) free(a.p);
        else (void)hipHostFree(a.p);
    }
    a.p = nullptr;
    a.cap = 0;
    a.pageable = false;
}

// Staging in page-locked memory (the copies then run at link speed and really are asynchronous).  Where the system
// will not lock that much (ulimit -l, cgroup), pageable memory does the same job slower: a large batch of decoders or
// streams must not fail for want of LOCKED memory when the memory itself is there.
int pinned_reserve(pzg_ctx *ctx, Pinned &a, size_t bytes)
{
    bytes = (bytes + 4095u) & ~(size_t)4095u;
    if (a.cap >= bytes) return PZG_RC_OK;
    pinned_release(a);
    hipError_t e = hipHostMalloc((void **)&a.p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        a.p = (uint8_t *)malloc(bytes);
        if (!a.p) {
            hip_fail(ctx, e, "hipHostMalloc");
            return PZG_RC_NO_MEMORY;
        }
        a.pageable = true;
    }
    a.cap = bytes;
    return PZG_RC_OK;
}

int lane_prepare(pzg_ctx *ctx, Lane &ln)
{
    if (ln.ready) return PZG_RC_OK;
    // The launch stream gets the highest priority: the runtime multiplexes a process' streams onto a few hardware queues, and a
    // launch stream that shares an in-order queue with a copy stream waits for whole 300 MiB downloads (measured: the same
    // pinned call 44 ms or 55 ms depending on which streams the process had created before); priority streams live on queues
    // of their own class -- so the three streams of a pipeline get three different priorities: launches highest, uploads normal,
    // downloads lowest.
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ln.s_k, hipStreamNonBlocking, greatest));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ln.s_up, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ln.s_dn, hipStreamNonBlocking, least));
    }
    for (int c = 0; c < NSLOT; ++c) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ln.ev_up[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ln.ev_k[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ln.ev_dn[c], hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipEventCreate(&ln.ev_t0));
    HIP_TRY(ctx, hipEventCreate(&ln.ev_t1));
    if (hipMalloc((void **)&ln.d_counter, 256) != hipSuccess) return PZG_RC_NO_MEMORY;
    ln.ready = true;
    return PZG_RC_OK;
}

void lane_destroy(Lane &ln)
{
    for (int c = 0; c < NSLOT; ++c) {
        for (Arena *a : {&ln.d_in[c], &ln.d_out[c], &ln.d_meta[c], &ln.d_gz[c], &ln.d_ord[c]})
            if (a->p) (void)hipFree(a->p);
        for (Pinned *p : {&ln.h_in[c], &ln.h_out[c], &ln.h_meta[c]}) pinned_release(*p);
        if (ln.ev_up[c]) (void)hipEventDestroy(ln.ev_up[c]);
        if (ln.ev_k[c]) (void)hipEventDestroy(ln.ev_k[c]);
        if (ln.ev_dn[c]) (void)hipEventDestroy(ln.ev_dn[c]);
    }
    if (ln.ev_t0) (void)hipEventDestroy(ln.ev_t0);
    if (ln.ev_t1) (void)hipEventDestroy(ln.ev_t1);
    if (ln.d_counter) (void)hipFree(ln.d_counter);
    if (ln.d_strip.p) (void)hipFree(ln.d_strip.p);
    if (ln.s_k) (void)hipStreamDestroy(ln.s_k);
    if (ln.s_up) (void)hipStreamDestroy(ln.s_up);
    if (ln.s_dn) (void)hipStreamDestroy(ln.s_dn);
}

// every lane stream of a lane quiet again (error paths: no copy that targets caller memory may stay in flight)
void lane_drain(Lane &ln)
{
    if (ln.s_up) (void)hipStreamSynchronize(ln.s_up);
    if (ln.s_k) (void)hipStreamSynchronize(ln.s_k);
    if (ln.s_dn) (void)hipStreamSynchronize(ln.s_dn);
}

int shard_create(pzg_ctx *ctx, int device, std::unique_ptr<Shard> &out)
{
    std::unique_ptr<Shard> sh(new (std::nothrow) Shard());
    if (!sh) return PZG_RC_NO_MEMORY;
    sh->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&sh->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&sh->ev0) != hipSuccess || hipEventCreate(&sh->ev1) != hipSuccess)
        return PZG_RC_NO_DEVICE;
    sh->stream = sh->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) sh->num_cus = prop.multiProcessorCount;
    if (hipMalloc((void **)&sh->d_counters, 256 * COUNTER_SLOTS) != hipSuccess) return PZG_RC_NO_MEMORY;
    for (int k = 0; k < STRIP_SLOTS; ++k)
        if (hipEventCreateWithFlags(&sh->ev_strip[k], hipEventDisableTiming) != hipSuccess) return PZG_RC_NO_DEVICE;
    if (hipEventCreateWithFlags(&sh->ev_bundle, hipEventDisableTiming) != hipSuccess) return PZG_RC_NO_DEVICE;
    if (hipHostMalloc((void **)&sh->h_bundle, 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        sh->h_bundle = nullptr;  // (tolerated: every launch probes)
    }
    out = std::move(sh);
    (void)ctx;
    return PZG_RC_OK;
}

void shard_destroy(Shard &sh)
{
    (void)hipSetDevice(sh.device);
    (void)hipDeviceSynchronize();
    for (Lane &ln : sh.lanes) lane_destroy(ln);
    for (Arena *a : {&sh.a_adler, &sh.a_scratch, &sh.a_dec})
        if (a->p) (void)hipFree(a->p);
    for (int k = 0; k < COUNTER_SLOTS; ++k)
        for (Arena *a : {&sh.a_gz[k], &sh.a_order[k]})
            if (a->p) (void)hipFree(a->p);
    for (int k = 0; k < STRIP_SLOTS; ++k) {
        if (sh.a_strip[k].p) (void)hipFree(sh.a_strip[k].p);
        if (sh.ev_strip[k]) (void)hipEventDestroy(sh.ev_strip[k]);
    }
    if (sh.h_bundle) (void)hipHostFree(sh.h_bundle);
    if (sh.ev_bundle) (void)hipEventDestroy(sh.ev_bundle);
    if (sh.d_counters) (void)hipFree(sh.d_counters);
    if (sh.ev0) (void)hipEventDestroy(sh.ev0);
    if (sh.ev1) (void)hipEventDestroy(sh.ev1);
    if (sh.own_stream) (void)hipStreamDestroy(sh.own_stream);
}

int ctx_create(const std::vector<int> &devices, pzg_ctx **out)
{
    pzg_ctx *ctx = new (std::nothrow) pzg_ctx();
    if (!ctx) return PZG_RC_NO_MEMORY;
    for (int d : devices) {
        std::unique_ptr<Shard> sh;
        int rc = shard_create(ctx, d, sh);
        if (rc != PZG_RC_OK) {
            if (sh) shard_destroy(*sh);
            for (auto &s : ctx->shards) shard_destroy(*s);
            delete ctx;
            return rc;
        }
        ctx->shards.push_back(std::move(sh));
    }
    if (const char *e = getenv("PZG_RING_BITS")) {  // environment override of the default size class
        const int rb = atoi(e);
        if (rb >= 11 && rb <= 15) ctx->ring_bits = rb;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    ctx->helpers.reset(new Helpers(hw > 24u ? 24u : hw < 1u ? 1u : hw));  // (PZG_OPT_HOST_THREADS changes it)
    *out = ctx;
    return PZG_RC_OK;
}

void ctx_unref(pzg_ctx *ctx)
{
    if (ctx->refs.fetch_sub(1) != 1) return;
    for (auto &s : ctx->shards) shard_destroy(*s);
    delete ctx;
}

inline bool ctx_live(const pzg_ctx *ctx) { return ctx && !ctx->closed.load(std::memory_order_acquire); }

// ---- the device-pointer path: everything already lives on shard 0's device -------------------------------------
int launch_device(pzg_ctx *ctx, Shard &sh, pzg::InflateArgs a, uint32_t flags)
{
    std::lock_guard<std::mutex> g(sh.mu);
    HIP_TRY(ctx, hipSetDevice(sh.device));
    const uint32_t slot = sh.next_counter.fetch_add(1u) % COUNTER_SLOTS;  // its own counter: launches on different streams may overlap
    a.counter = sh.d_counters + 64u * slot;
    // (arena_reserve only ever frees a slot's old array through hipFree, which waits for the device: a launch still
    // reading it has finished by then)
    if (flags & PZG_GZIP) {
        int rc = arena_reserve(ctx, sh.a_gz[slot], 8 * (size_t)a.n);
        if (rc != PZG_RC_OK) return rc;
        a.gzip = 1;
        a.gz_expect = (uint32_t *)sh.a_gz[slot].p;
    }
    if (flags & PZG_LPT_ORDER) {  // longest streams first: a launch permutation built on the device from out_cap[]
        int rc = arena_reserve(ctx, sh.a_order[slot], 4 * (size_t)a.n + 1024);
        if (rc != PZG_RC_OK) return rc;
        uint32_t *ord = (uint32_t *)sh.a_order[slot].p;
        HIP_TRY(ctx, pzg::launch_order(a.out_cap, a.n, ord + 256, ord, sh.stream));
        a.order = ord + 256;
    }
#if defined(PZG_PROFILE)
    a.prof_out = (uint64_t *)ctx->prof_buf;
#endif
    // the kernels' token scratch: the next of STRIP_SLOTS arenas, after the launch that used it last (no scratch, no strips:
    // the kernels then decode by windows alone -- slower, never wrong)
    const uint32_t ss = sh.next_strip++ % STRIP_SLOTS;
    if (sh.strip_used[ss]) HIP_TRY(ctx, hipStreamWaitEvent(sh.stream, sh.ev_strip[ss], 0));
    strip_for_launch(ctx, sh.a_strip[ss], sh.num_cus, a.n, a.gzip, a);
#if defined(PZG_LAB)  // lab builds only: the windows alone
    if (getenv("PZG_NO_STRIPS")) a.strip = nullptr;
#endif
    if (a.strip && sh.a_strip[ss].prof_off != (ctx->profile.load() == 0)) {  // PZG_OPT_PROFILE changed since this scratch was last used
        const bool off = ctx->profile.load() == 0;
        HIP_TRY(ctx, pzg::launch_profile_switch(a.strip, (uint32_t)(sh.a_strip[ss].cap / pzg::inflate_strip_wave_bytes()), off, sh.stream));
        sh.a_strip[ss].prof_off = off;
    }
    a.bundle = (ctx->bundles.load() == 2 || (ctx->bundles.load() == 1 && a.n >= BUNDLE_MIN_STREAMS)) && !(flags & PZG_GZIP) && !a.dict_len ? 1u : 0u;
    if (a.bundle && ctx->bundles.load() == 1 && sh.h_bundle) {
        // (the probe costs a launch of dynamic-code streams ~1 %: the bundle kernel, and two words every stream-wave reads first)
        if (sh.bundle_pending && hipEventQuery(sh.ev_bundle) == hipSuccess) {
            sh.bundle_pending = false;
            if (sh.h_bundle[0] == 1u) {  // (the count + 1: see pzg_inflate_kernel.h)
                sh.bundle_backoff = sh.bundle_backoff >= 31u ? 64u : 2u * sh.bundle_backoff + 2u;
                sh.bundle_skip = sh.bundle_backoff;
            } else {
                sh.bundle_backoff = 0u;
            }
        }
        if (sh.bundle_skip != 0u) {
            sh.bundle_skip -= 1u;
            a.bundle = 0u;
        } else if (!sh.bundle_pending) {
            sh.h_bundle[0] = 0u;
            a.bundle_report = sh.h_bundle;  // (page-locked host memory: the device writes it over the link)
        }
    }
    HIP_TRY(ctx, hipEventRecord(sh.ev0, sh.stream));
    HIP_TRY(ctx, pzg::launch_inflate(a, ctx->ring_bits, sh.num_cus, sh.stream));
    HIP_TRY(ctx, hipEventRecord(sh.ev1, sh.stream));
    HIP_TRY(ctx, hipEventRecord(sh.ev_strip[ss], sh.stream));
    if (a.bundle_report) {  // what the probe found (written by the launch's last kernel), for the launches to come
        HIP_TRY(ctx, hipEventRecord(sh.ev_bundle, sh.stream));
        sh.bundle_pending = true;
    }
    sh.strip_used[ss] = true;
    sh.timed = true;
    sh.last_was_host = false;
    if (!(flags & PZG_ASYNC)) HIP_TRY(ctx, hipStreamSynchronize(sh.stream));
    return PZG_RC_OK;
}

// ---- the host-pointer path on one shard, for the streams idx[0..m) of the caller's batch -----------------------
struct HostBatch {
    const uint8_t *in_base;
    const uint64_t *in_off, *in_len;
    uint8_t *out_base;
    const uint64_t *out_off, *out_cap;
    uint64_t *out_len;
    int32_t *status;
    uint32_t *detail;
    uint64_t *in_used;
    uint32_t *adler;
    uint32_t flags;
};

struct Range {
    uint32_t lo, hi;          // positions in idx[]
    size_t in_bytes, out_bytes;  // packed sizes (PZG_HOST_PINNED: the spans of the caller's arenas the range covers)
    uint64_t in_lo = 0, out_lo = 0;  // PZG_HOST_PINNED: where those spans start in the caller's arenas
    bool mixed = false;       // PZG_HOST_PINNED: capacities differ inside the range (launched longest first by a device-side permutation)
};

inline size_t pad16(size_t x) { return (x + 15u) & ~(size_t)15u; }
inline size_t pad256(size_t x) { return (x + 255u) & ~(size_t)255u; }

// The host-pointer path.  Two forms:
//   staged (default)    pageable caller memory: the streams are packed into page-locked staging (no gaps travel), launched
//                       longest first, and the decoded bytes come back through page-locked staging and are copied out;
//   PZG_HOST_PINNED     the caller's arenas are page-locked (pzg_host_alloc) and laid out in ascending order: the copy
//                       engines read and write them directly -- no packing, no staging, no copy-out.  The device holds a
//                       mirror of the spans the batch covers; a range uploads its span of the input arena and downloads its
//                       span of the output arena (gaps between extents travel along).
int host_path(pzg_ctx *ctx, Shard &sh, const HostBatch &b, const uint32_t *idx, uint32_t m)
{
    if (m == 0) return PZG_RC_OK;
    const bool pinned = (b.flags & PZG_HOST_PINNED) != 0;
    // the first free lane of this shard in a fixed order (a lone caller always lands on the same, warm one); if all
    // are taken, queue on one of them
    Lane *lane = nullptr;
    std::unique_lock<std::mutex> lk;
    for (int t = 0; t < LANES_PER_SHARD && !lane; ++t) {
        std::unique_lock<std::mutex> trial(sh.lanes[t].mu, std::try_to_lock);
        if (trial.owns_lock()) {
            lane = &sh.lanes[t];
            lk = std::move(trial);
        }
    }
    if (!lane) {
        lane = &sh.lanes[sh.next_lane.fetch_add(1u) % LANES_PER_SHARD];
        lk = std::unique_lock<std::mutex>(lane->mu);
    }
    Lane &ln = *lane;
    HIP_TRY(ctx, hipSetDevice(sh.device));
    int rc = lane_prepare(ctx, ln);
    if (rc != PZG_RC_OK) return rc;
#if defined(PZG_LAB)
    const bool trace = getenv("PZG_TRACE_HOST") != nullptr;
#else
    const bool trace = false;
#endif
    const auto t_call0 = std::chrono::steady_clock::now();

    // ranges of the (already ordered) stream list: a few per call, each below RANGE_MAX_OUT of packed output
    size_t tot_in = 0, tot_out = 0;
    uint64_t g_in_lo = 0, g_out_lo = 0;  // PZG_HOST_PINNED: the batch's spans start here (extents ascend: validated by the caller)
    if (pinned) {
        g_in_lo = b.in_off[idx[0]];
        g_out_lo = b.out_off[idx[0]];
        tot_in = (size_t)(b.in_off[idx[m - 1]] + b.in_len[idx[m - 1]] - g_in_lo);
        tot_out = (size_t)(b.out_off[idx[m - 1]] + b.out_cap[idx[m - 1]] - g_out_lo);
    } else {
        for (uint32_t k = 0; k < m; ++k) {
            tot_in += pad16(b.in_len[idx[k]]);
            tot_out += pad16(b.out_cap[idx[k]]);
        }
    }
    size_t target_out = tot_out / 6 + 1, target_in = tot_in / 6 + 1;
    if (tot_in + tot_out < (96ull << 20) || m < 2048u) target_out = target_in = ~(size_t)0 >> 1;  // small batch: one range
    if (!pinned && target_out > RANGE_MAX_OUT) target_out = RANGE_MAX_OUT;
#if defined(PZG_LAB)
    if (const char *e = getenv("PZG_HOST_RANGES")) {  // experiment knob
        const int v = atoi(e);
        if (v >= 1) {
            target_out = tot_out / (size_t)v + 1;
            target_in = tot_in / (size_t)v + 1;
        }
    }
#endif
    std::vector<Range> rg;
    {
        Range r{0, 0, 0, 0};
        for (uint32_t k = 0; k < m; ++k) {
            const size_t ib = pad16(b.in_len[idx[k]]), ob = pad16(b.out_cap[idx[k]]);
            if (r.hi > r.lo && (r.out_bytes + ob > target_out || r.in_bytes + ib > target_in)) {
                rg.push_back(r);
                r = Range{k, k, 0, 0};
            }
            r.hi = k + 1;
            r.in_bytes += ib;
            r.out_bytes += ob;
        }
        rg.push_back(r);
    }
    if (pinned) {
        for (Range &r : rg) {
            const uint32_t f = idx[r.lo], l = idx[r.hi - 1];
            r.in_lo = b.in_off[f];
            r.out_lo = b.out_off[f];
            r.in_bytes = (size_t)(b.in_off[l] + b.in_len[l] - r.in_lo);
            r.out_bytes = (size_t)(b.out_off[l] + b.out_cap[l] - r.out_lo);
            for (uint32_t k = r.lo + 1; k < r.hi && !r.mixed; ++k) r.mixed = b.out_cap[idx[k]] != b.out_cap[f];
        }
    }
    size_t max_in = 0, max_out = 0, max_n = 0;
    for (const Range &r : rg) {
        max_in = std::max(max_in, r.in_bytes);
        max_out = std::max(max_out, r.out_bytes);
        max_n = std::max(max_n, (size_t)(r.hi - r.lo));
    }
    // per-range meta block (u64[n] x 6 | i32 status | u32 adler | u32 detail[2n]), identical on host and device
    const size_t meta_bytes = 64 * max_n;
    const int nslot = (int)std::min<size_t>(rg.size(), (size_t)NSLOT);
    if (pinned) {  // one mirror of each arena's span for the whole call
        if ((rc = arena_reserve(ctx, ln.d_in[0], tot_in + 64)) != PZG_RC_OK) return rc;
        if ((rc = arena_reserve(ctx, ln.d_out[0], tot_out + 64)) != PZG_RC_OK) return rc;
    }
    for (int s = 0; s < nslot; ++s) {
        if (!pinned) {
            if ((rc = arena_reserve(ctx, ln.d_in[s], max_in + 64)) != PZG_RC_OK) return rc;
            if ((rc = arena_reserve(ctx, ln.d_out[s], max_out + 64)) != PZG_RC_OK) return rc;
            if ((rc = pinned_reserve(ctx, ln.h_in[s], max_in + 64)) != PZG_RC_OK) return rc;
            if ((rc = pinned_reserve(ctx, ln.h_out[s], max_out + 64)) != PZG_RC_OK) return rc;
        } else if ((rc = arena_reserve(ctx, ln.d_ord[s], 4 * max_n + 1024)) != PZG_RC_OK) {
            return rc;
        }
        if ((rc = arena_reserve(ctx, ln.d_meta[s], meta_bytes + 64)) != PZG_RC_OK) return rc;
        if ((b.flags & PZG_GZIP) && (rc = arena_reserve(ctx, ln.d_gz[s], 8 * max_n + 64)) != PZG_RC_OK) return rc;
        if ((rc = pinned_reserve(ctx, ln.h_meta[s], meta_bytes + 64)) != PZG_RC_OK) return rc;
    }
    pzg::InflateArgs strip_args{};  // (all launches of the call go to s_k, one after the other: one scratch)
    strip_for_launch(ctx, ln.d_strip, sh.num_cus, (uint32_t)max_n, (b.flags & PZG_GZIP) ? 1u : 0u, strip_args);
    const unsigned helpers = (!pinned && (tot_in + tot_out) >= (32ull << 20)) ? ctx->helpers->size() : 1u;

    auto meta_ptrs = [&](uint8_t *base, size_t nn, uint64_t *&ioff, uint64_t *&ilen, uint64_t *&ooff, uint64_t *&ocap, uint64_t *&olen,
                         uint64_t *&used, int32_t *&st, uint32_t *&ad, uint32_t *&det) {
        ioff = (uint64_t *)base;
        ilen = ioff + nn;
        ooff = ilen + nn;
        ocap = ooff + nn;
        olen = ocap + nn;
        used = olen + nn;
        st = (int32_t *)(used + nn);
        ad = (uint32_t *)(st + nn);
        det = ad + nn;
    };

    // lay out the range's extents; staged: pack its streams into the pinned input slot
    auto pack = [&](const Range &r, int s) {
        const size_t nn = r.hi - r.lo;
        uint64_t *ioff, *ilen, *ooff, *ocap, *olen, *used;
        int32_t *st;
        uint32_t *ad, *det;
        meta_ptrs(ln.h_meta[s].p, nn, ioff, ilen, ooff, ocap, olen, used, st, ad, det);
        if (pinned) {  // the caller's own layout, relative to the start of the batch's spans
            for (size_t q = 0; q < nn; ++q) {
                const uint32_t i = idx[r.lo + q];
                ioff[q] = b.in_off[i] - g_in_lo;
                ilen[q] = b.in_len[i];
                ooff[q] = b.out_off[i] - g_out_lo;
                ocap[q] = b.out_cap[i];
            }
            return;
        }
        size_t ip = 0, op = 0;
        for (size_t q = 0; q < nn; ++q) {
            const uint32_t i = idx[r.lo + q];
            ioff[q] = ip;
            ilen[q] = b.in_len[i];
            ooff[q] = op;
            ocap[q] = b.out_cap[i];
            ip += pad16(b.in_len[i]);
            op += pad16(b.out_cap[i]);
        }
        uint8_t *dst = ln.h_in[s].p;
        ctx->helpers->run(helpers, [&](unsigned t, unsigned parts) {
            const size_t q0 = nn * t / parts, q1 = nn * (t + 1) / parts;
            for (size_t q = q0; q < q1; ++q) {
                const uint32_t i = idx[r.lo + q];
                if (b.in_len[i]) memcpy(dst + ioff[q], b.in_base + b.in_off[i], b.in_len[i]);
            }
        });
    };
    // hand a finished range to the caller: results, and (staged) the decoded bytes out of the pinned output slot
    auto unpack = [&](const Range &r, int s) {
        const size_t nn = r.hi - r.lo;
        uint64_t *ioff, *ilen, *ooff, *ocap, *olen, *used;
        int32_t *st;
        uint32_t *ad, *det;
        meta_ptrs(ln.h_meta[s].p, nn, ioff, ilen, ooff, ocap, olen, used, st, ad, det);
        const uint8_t *src = pinned ? nullptr : ln.h_out[s].p;
        ctx->helpers->run(helpers, [&](unsigned t, unsigned parts) {
            const size_t q0 = nn * t / parts, q1 = nn * (t + 1) / parts;
            for (size_t q = q0; q < q1; ++q) {
                const uint32_t i = idx[r.lo + q];
                b.out_len[i] = olen[q];
                b.status[i] = st[q];
                if (b.in_used) b.in_used[i] = used[q];
                if (b.adler) b.adler[i] = ad[q];
                if (b.detail) {
                    b.detail[2 * (size_t)i] = det[2 * q];
                    b.detail[2 * (size_t)i + 1] = det[2 * q + 1];
                }
                const uint64_t nb = olen[q] < ocap[q] ? olen[q] : ocap[q];
                if (src && nb) memcpy(b.out_base + b.out_off[i], src + ooff[q], nb);
            }
        });
    };

    hipError_t herr = hipSuccess;
    const char *hwhat = "";
    double t_pack = 0, t_unpack = 0, t_wait = 0;
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
#define LANE_TRY(call)                 \
    do {                               \
        if (herr == hipSuccess) {      \
            herr = (call);             \
            if (herr != hipSuccess) hwhat = #call; \
        }                              \
    } while (0)

    // Two host threads per call: this one packs and issues range after range; a second one waits for each range's
    // download and copies it out to the caller.  A slot is reused once its previous occupant has been copied out.
    const size_t R = rg.size();
    // a one-range call keeps its three steps on ONE stream (no cross-stream events; concurrent small calls from other
    // lanes then sit on different hardware queues and overlap); bigger ones overlap their own ranges on three streams
    const hipStream_t s_up = R > 1 ? ln.s_up : ln.s_k, s_dn = R > 1 ? ln.s_dn : ln.s_k;
    std::mutex pm;
    std::condition_variable pcv;
    size_t issued = 0, unpacked = 0;  // ranges issued / handed back so far
    bool abort_drain = false;
    hipError_t derr = hipSuccess;
    // one range handed back: wait for its download, copy it out (a std::bad_alloc of the helpers' job list is reported, not thrown
    // across a thread boundary)
    auto drain_one = [&](size_t c) {
        const int s = (int)(c % nslot);
        auto tw = tick();
        hipError_t e = hipEventSynchronize(ln.ev_dn[s]);
        t_wait += since(tw);
        auto tu = tick();
        if (e == hipSuccess) {
            try {
                unpack(rg[c], s);
            } catch (...) {
                e = hipErrorOutOfMemory;
            }
        }
        t_unpack += since(tu);
        {
            std::lock_guard<std::mutex> g(pm);
            if (e != hipSuccess && derr == hipSuccess) derr = e;
            unpacked = c + 1;
        }
        pcv.notify_all();
    };
    auto drain = [&] {
        (void)hipSetDevice(sh.device);
        for (size_t c = 0; c < R; ++c) {
            {
                std::unique_lock<std::mutex> g(pm);
                pcv.wait(g, [&] { return issued > c || abort_drain; });
                if (abort_drain && issued <= c) return;
            }
            drain_one(c);
        }
    };
    CallThread drainer;
    drainer.wake = [&] {
        {
            std::lock_guard<std::mutex> g(pm);
            abort_drain = true;
        }
        pcv.notify_all();
    };
    const bool threaded = R > 1 && drainer.start(drain);  // (no thread to be had: this thread hands the ranges back itself)
    size_t drained_inline = 0;
    for (size_t c = 0; c < R && herr == hipSuccess; ++c) {
        const Range &r = rg[c];
        const int s = (int)(c % nslot);
        const size_t nn = r.hi - r.lo;
        if (c >= (size_t)nslot) {  // the slot's previous occupant (range c - nslot) must have left the pinned buffers
            if (!threaded)
                while (drained_inline + nslot <= c) drain_one(drained_inline++);
            std::unique_lock<std::mutex> g(pm);
            pcv.wait(g, [&] { return unpacked + nslot > c; });
            if (derr != hipSuccess) {
                herr = derr;
                hwhat = "hipEventSynchronize (download)";
                break;
            }
        }
        auto tp = tick();
        pack(r, s);
        t_pack += since(tp);
        uint8_t *dm = (uint8_t *)ln.d_meta[s].p;
        uint8_t *din = (uint8_t *)ln.d_in[pinned ? 0 : s].p, *dout = (uint8_t *)ln.d_out[pinned ? 0 : s].p;
        LANE_TRY(hipMemcpyAsync(dm, ln.h_meta[s].p, 32 * nn, hipMemcpyHostToDevice, s_up));  // the four extent arrays
        if (r.in_bytes) {
            if (pinned) LANE_TRY(hipMemcpyAsync(din + (r.in_lo - g_in_lo), b.in_base + r.in_lo, r.in_bytes, hipMemcpyHostToDevice, s_up));
            else LANE_TRY(hipMemcpyAsync(din, ln.h_in[s].p, r.in_bytes, hipMemcpyHostToDevice, s_up));
        }
        if (R > 1) {
            LANE_TRY(hipEventRecord(ln.ev_up[s], s_up));
            LANE_TRY(hipStreamWaitEvent(ln.s_k, ln.ev_up[s], 0));
        }
        pzg::InflateArgs a{};
        uint64_t *ioff, *ilen, *ooff, *ocap, *olen, *used;
        int32_t *st;
        uint32_t *ad, *det;
        meta_ptrs(dm, nn, ioff, ilen, ooff, ocap, olen, used, st, ad, det);
        a.in_base = din;
        a.out_base = dout;
        a.in_off = ioff;
        a.in_len = ilen;
        a.out_off = ooff;
        a.out_cap = ocap;
        a.out_len = olen;
        a.in_used = used;
        a.status = st;
        a.adler = ad;
        a.detail = det;
        a.n = (uint32_t)nn;
        a.counter = ln.d_counter;
        a.strip = strip_args.strip;
        a.strip_waves = strip_args.strip_waves;
        if (b.flags & PZG_GZIP) {
            a.gzip = 1;
            a.gz_expect = (uint32_t *)ln.d_gz[s].p;
        }
#if defined(PZG_PROFILE)
        a.prof_out = ctx->prof_buf ? (uint64_t *)ctx->prof_buf + 16 * (size_t)r.lo : nullptr;
#endif
        if (c == 0) LANE_TRY(hipEventRecord(ln.ev_t0, ln.s_k));
        if (pinned && r.mixed) {  // the caller's order stays: the longest streams are launched first through a permutation
            uint32_t *ord = (uint32_t *)ln.d_ord[s].p;
            LANE_TRY(pzg::launch_order(ocap, a.n, ord + 256, ord, ln.s_k));
            a.order = ord + 256;
        }
        LANE_TRY(pzg::launch_inflate(a, ctx->ring_bits, sh.num_cus, ln.s_k));
        if (c + 1 == R) LANE_TRY(hipEventRecord(ln.ev_t1, ln.s_k));
        if (R > 1) {
            LANE_TRY(hipEventRecord(ln.ev_k[s], ln.s_k));
            LANE_TRY(hipStreamWaitEvent(s_dn, ln.ev_k[s], 0));
        }
        LANE_TRY(hipMemcpyAsync(ln.h_meta[s].p + 32 * nn, dm + 32 * nn, 32 * nn, hipMemcpyDeviceToHost, s_dn));  // results
        if (r.out_bytes) {
            if (pinned) LANE_TRY(hipMemcpyAsync(b.out_base + r.out_lo, dout + (r.out_lo - g_out_lo), r.out_bytes, hipMemcpyDeviceToHost, s_dn));
            else LANE_TRY(hipMemcpyAsync(ln.h_out[s].p, dout, r.out_bytes, hipMemcpyDeviceToHost, s_dn));
        }
        LANE_TRY(hipEventRecord(ln.ev_dn[s], s_dn));
        if (herr == hipSuccess) {
            {
                std::lock_guard<std::mutex> g(pm);
                issued = c + 1;
            }
            pcv.notify_all();
        }
    }
    if (threaded) {
        drainer.wake();  // (a no-op when every range was issued: the drainer finishes them all first)
        drainer.join();
    } else if (herr == hipSuccess) {
        while (drained_inline < R) drain_one(drained_inline++);
    }
    if (herr == hipSuccess && derr != hipSuccess) {
        herr = derr;
        hwhat = derr == hipErrorOutOfMemory ? "copy-out of a range (out of memory)" : "hipEventSynchronize (download)";
    }
#undef LANE_TRY
    if (herr != hipSuccess) {
        lane_drain(ln);
        return hip_fail(ctx, herr, hwhat);
    }
    {   // the call's kernel span, measured by the call itself on its own lane's events (pzg_last_kernel_ms): calls of
        // other threads on other lanes cannot get between the two
        float ms = -1.0f;
        const bool ok = hipEventSynchronize(ln.ev_t1) == hipSuccess && hipEventElapsedTime(&ms, ln.ev_t0, ln.ev_t1) == hipSuccess;
        std::lock_guard<std::mutex> g(sh.mu);
        sh.host_ms = ok ? (double)ms : -1.0;
        sh.last_was_host = true;
        sh.timed = true;
    }
    if (trace)
        fprintf(stderr, "[pzg] host path%s: %u streams, %.1f MiB in, %.1f MiB out, %zu range(s), %u helper thread(s): %.1f ms "
                "(packing %.1f on the issuing thread; on the draining thread: waiting for downloads %.1f, copy-out %.1f)\n", pinned ? " (pinned arenas)" : "",
                m, tot_in / 1048576.0, tot_out / 1048576.0, R, helpers,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call0).count(), t_pack, t_wait, t_unpack);
    return PZG_RC_OK;
}

// ---- preset dictionaries (extension): a plain synchronous path on lane 0 -- pack, upload, launch, download ----------
int dict_path(pzg_ctx *ctx, Shard &sh, const HostBatch &b, const uint8_t *dict_base, const uint64_t *dict_off, const uint64_t *dict_len,
              uint32_t n)
{
    Lane &ln = sh.lanes[0];
    std::lock_guard<std::mutex> lk(ln.mu);
    HIP_TRY(ctx, hipSetDevice(sh.device));
    int rc = lane_prepare(ctx, ln);
    if (rc != PZG_RC_OK) return rc;
    std::vector<uint64_t> meta(6 * (size_t)n);  // in_off | in_len | out_off | out_cap | dict_off | dict_len, packed layout
    uint64_t *ioff = meta.data(), *ilen = ioff + n, *ooff = ilen + n, *ocap = ooff + n, *doff = ocap + n, *dlen = doff + n;
    size_t ip = 0, op = 0;
    for (uint32_t i = 0; i < n; ++i) {
        ioff[i] = ip;
        ilen[i] = b.in_len[i];
        ip += pad16(b.in_len[i]);
        doff[i] = ip;
        dlen[i] = dict_len[i];
        ip += pad16(dict_len[i]);
        ooff[i] = op;
        ocap[i] = b.out_cap[i];
        op += pad16(b.out_cap[i]);
    }
    std::vector<uint8_t> hin(ip + 16);
    for (uint32_t i = 0; i < n; ++i) {
        if (ilen[i]) memcpy(hin.data() + ioff[i], b.in_base + b.in_off[i], ilen[i]);
        if (dlen[i]) memcpy(hin.data() + doff[i], dict_base + dict_off[i], dlen[i]);
    }
    if ((rc = arena_reserve(ctx, ln.d_in[0], ip + 64)) != PZG_RC_OK) return rc;
    if ((rc = arena_reserve(ctx, ln.d_out[0], op + 64)) != PZG_RC_OK) return rc;
    if ((rc = arena_reserve(ctx, ln.d_meta[0], 80 * (size_t)n + 64)) != PZG_RC_OK) return rc;
    uint8_t *dm = (uint8_t *)ln.d_meta[0].p;
    hipStream_t st = ln.s_k;
    HIP_TRY(ctx, hipMemcpyAsync(dm, meta.data(), 48 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ln.d_in[0].p, hin.data(), ip, hipMemcpyHostToDevice, st));
    pzg::InflateArgs a{};
    a.in_base = (const uint8_t *)ln.d_in[0].p;
    a.dict_base = a.in_base;
    a.out_base = (uint8_t *)ln.d_out[0].p;
    a.in_off = (const uint64_t *)dm;
    a.in_len = a.in_off + n;
    a.out_off = a.in_len + n;
    a.out_cap = a.out_off + n;
    a.dict_off = a.out_cap + n;
    a.dict_len = a.dict_off + n;
    a.out_len = (uint64_t *)(dm + 48 * (size_t)n);
    a.in_used = a.out_len + n;
    a.status = (int32_t *)(a.in_used + n);
    a.adler = (uint32_t *)(a.status + n);
    a.detail = a.adler + n;
    a.n = n;
    a.counter = ln.d_counter;
    strip_for_launch(ctx, ln.d_strip, sh.num_cus, n, 0u, a);
    HIP_TRY(ctx, pzg::launch_inflate(a, ctx->ring_bits, sh.num_cus, st));
    std::vector<uint8_t> res(32 * (size_t)n), hout(op + 16);
    HIP_TRY(ctx, hipMemcpyAsync(res.data(), dm + 48 * (size_t)n, 32 * (size_t)n, hipMemcpyDeviceToHost, st));
    if (op) HIP_TRY(ctx, hipMemcpyAsync(hout.data(), ln.d_out[0].p, op, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const uint64_t *olen = (const uint64_t *)res.data(), *used = olen + n;
    const int32_t *stt = (const int32_t *)(used + n);
    const uint32_t *ad = (const uint32_t *)(stt + n), *det = ad + n;
    for (uint32_t i = 0; i < n; ++i) {
        b.out_len[i] = olen[i];
        b.status[i] = stt[i];
        if (b.in_used) b.in_used[i] = used[i];
        if (b.adler) b.adler[i] = ad[i];
        if (b.detail) {
            b.detail[2 * (size_t)i] = det[2 * (size_t)i];
            b.detail[2 * (size_t)i + 1] = det[2 * (size_t)i + 1];
        }
        const uint64_t nb = olen[i] < ocap[i] ? olen[i] : ocap[i];
        if (nb) memcpy(b.out_base + b.out_off[i], hout.data() + ooff[i], nb);
    }
    return PZG_RC_OK;
}

// longest first (by capacity): the launch order inside a shard; stable, so equal streams keep the caller's order
void lpt_order(const uint64_t *out_cap, std::vector<uint32_t> &idx)
{
    bool uniform = true;
    for (size_t k = 1; k < idx.size() && uniform; ++k) uniform = out_cap[idx[k]] == out_cap[idx[0]];
    if (!uniform) std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return out_cap[a] > out_cap[b]; });
}

}  // namespace

extern "C" {

#if defined(PZG_PROFILE)
// diagnostic builds only: device buffer of 16 uint64 per stream the kernel fills with cycle counters
PZG_API int pzg_prof_buffer(pzg_ctx *ctx, uint32_t n, uint64_t *host_out)
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
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return PZG_RC_NO_DEVICE;
        if (device < 0 || device >= ndev) return PZG_RC_BAD_ARG;
        return ctx_create({device}, out);
    } catch (...) {
        return PZG_RC_NO_MEMORY;
    }
}

int pzg_init_mask(uint32_t device_mask, pzg_ctx **out)
{
    if (!out) return PZG_RC_BAD_ARG;
    *out = nullptr;
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return PZG_RC_NO_DEVICE;
        if (device_mask == 0) device_mask = ndev >= 32 ? ~0u : (1u << ndev) - 1u;  // 0 = every visible device
        std::vector<int> devs;
        for (int d = 0; d < 32; ++d)
            if (device_mask & (1u << d)) {
                if (d >= ndev) return PZG_RC_BAD_ARG;
                devs.push_back(d);
            }
        if (devs.empty()) return PZG_RC_BAD_ARG;
        return ctx_create(devs, out);
    } catch (...) {
        return PZG_RC_NO_MEMORY;
    }
}

int pzg_init_devices(const int32_t *devices, uint32_t ndevices, pzg_ctx **out)
{
    if (!out) return PZG_RC_BAD_ARG;
    *out = nullptr;
    if (!devices || ndevices == 0 || ndevices > 64u) return PZG_RC_BAD_ARG;
    try {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return PZG_RC_NO_DEVICE;
        std::vector<int> devs;
        for (uint32_t k = 0; k < ndevices; ++k) {
            if (devices[k] < 0 || devices[k] >= ndev) return PZG_RC_BAD_ARG;
            devs.push_back((int)devices[k]);  // (a device may be named more than once: one shard per entry)
        }
        return ctx_create(devs, out);
    } catch (...) {
        return PZG_RC_NO_MEMORY;
    }
}

void *pzg_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 1;
    // portable: page-locked for every device of the node (a multi-device context's copy engines all read / write it)
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void pzg_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int pzg_device_count(pzg_ctx *ctx) { return ctx_live(ctx) ? (int)ctx->shards.size() : 0; }

void pzg_shutdown(pzg_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->closed.exchange(true)) return;  // (a second shutdown while decoders keep the context alive: nothing to drop)
    // quiet before the handle goes: nothing enqueued through it may still be running on caller memory
    // (device-wide, not per stream: a stream borrowed through pzg_set_stream may already have been destroyed by its owner
    // -- finalisers run in any order -- and its handle must not be touched; whatever was enqueued on it is waited for here)
    for (auto &s : ctx->shards) {
        std::lock_guard<std::mutex> g(s->mu);
        if (hipSetDevice(s->device) == hipSuccess) (void)hipDeviceSynchronize();
        s->stream = s->own_stream;
    }
    ctx_unref(ctx);  // live decoders hold their own references: the last pzg_decoder_destroy frees the context
}

int pzg_set_stream(pzg_ctx *ctx, void *hip_stream)
{
    if (!ctx_live(ctx)) return PZG_RC_BAD_ARG;
    Shard &sh = *ctx->shards[0];
    std::lock_guard<std::mutex> g(sh.mu);
    sh.stream = (hipStream_t)hip_stream;  // NULL = the default (null) stream
    return PZG_RC_OK;
}

int pzg_reset_stream(pzg_ctx *ctx)
{
    if (!ctx_live(ctx)) return PZG_RC_BAD_ARG;
    Shard &sh = *ctx->shards[0];
    std::lock_guard<std::mutex> g(sh.mu);
    sh.stream = sh.own_stream;
    return PZG_RC_OK;
}

int pzg_set_option(pzg_ctx *ctx, int option, int64_t value)
{
    if (!ctx_live(ctx)) return PZG_RC_BAD_ARG;
    if (option == PZG_OPT_RING_BITS && value >= 11 && value <= 15) {
        ctx->ring_bits = (int)value;
        return PZG_RC_OK;
    }
    if (option == PZG_OPT_SCRATCH_BYTES && value >= 0) {  // (takes effect launch by launch: an arena larger than its share is given back when next used)
        ctx->scratch_cap.store((uint64_t)value);
        return PZG_RC_OK;
    }
    if (option == PZG_OPT_PROFILE && (value == 0 || value == 1)) {
        ctx->profile.store((int)value);
        return PZG_RC_OK;
    }
    if (option == PZG_OPT_BUNDLES && value >= 0 && value <= 2) {
        ctx->bundles.store((int)value);
        // (setting it -- to the value it has, too -- makes the next launch look again: a caller that knows its next batches are of
        // another kind need not wait out the launches a context goes without looking after batches that had nothing for the bundles)
        for (auto &s : ctx->shards) {
            std::lock_guard<std::mutex> g(s->mu);
            s->bundle_skip = 0u;
            s->bundle_backoff = 0u;
        }
        return PZG_RC_OK;
    }
    if (option == PZG_OPT_HOST_THREADS && value >= 1 && value <= 256) {  // (not while host-pointer calls are running)
        try {
            ctx->helpers.reset(new Helpers((unsigned)value));
        } catch (...) {
            return PZG_RC_NO_MEMORY;
        }
        return PZG_RC_OK;
    }
    return PZG_RC_BAD_ARG;
}

int pzg_sync(pzg_ctx *ctx)
{
    if (!ctx_live(ctx)) return PZG_RC_BAD_ARG;
    for (auto &s : ctx->shards) {
        std::lock_guard<std::mutex> g(s->mu);
        HIP_TRY(ctx, hipSetDevice(s->device));
        HIP_TRY(ctx, hipStreamSynchronize(s->stream));
    }
    return PZG_RC_OK;
}

int pzg_decompress_many(pzg_ctx *ctx, const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                        uint8_t *out_base, const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                        int32_t *status, uint32_t *detail, uint64_t *in_used, uint32_t *adler, uint32_t n,
                        uint32_t flags)
{
    return pzg_decompress_many_dict(ctx, in_base, in_off, in_len, nullptr, nullptr, nullptr, out_base, out_off, out_cap, out_len, status,
                                    detail, in_used, adler, n, flags);
}

int pzg_decompress_many_dict(pzg_ctx *ctx, const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                             const uint8_t *dict_base, const uint64_t *dict_off, const uint64_t *dict_len, uint8_t *out_base,
                             const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len, int32_t *status, uint32_t *detail,
                             uint64_t *in_used, uint32_t *adler, uint32_t n, uint32_t flags)
{
    if (!ctx_live(ctx)) return PZG_RC_BAD_ARG;
    const bool with_dict = dict_base && dict_off && dict_len;
    if (with_dict && (flags & PZG_GZIP)) return PZG_RC_BAD_ARG;  // (preset dictionaries are a zlib-container notion)
    if (n == 0) return PZG_RC_OK;
    if (!in_base || !in_off || !in_len || !out_off || !out_cap || !out_len || !status) return PZG_RC_BAD_ARG;
    if ((flags & PZG_ASYNC) && !(flags & PZG_DEVICE_PTRS)) return PZG_RC_BAD_ARG;
    try {
        if (flags & PZG_DEVICE_PTRS) {
            if (!out_base || ctx->shards.size() != 1) return PZG_RC_BAD_ARG;  // device pointers belong to ONE device
            pzg::InflateArgs a{};
            a.in_base = in_base;
            a.in_off = in_off;
            a.in_len = in_len;
            a.out_base = out_base;
            a.out_off = out_off;
            a.out_cap = out_cap;
            a.out_len = out_len;
            a.status = status;
            a.detail = detail;
            a.in_used = in_used;
            a.adler = adler;
            a.n = n;
            if (with_dict) {
                a.dict_base = dict_base;
                a.dict_off = dict_off;
                a.dict_len = dict_len;
            }
            return launch_device(ctx, *ctx->shards[0], a, flags);
        }
        // host pointers: validate the extents (a wrapped offset + length would size an arena far too small)
        bool any_out = false;
        for (uint32_t i = 0; i < n; ++i) {
            if (in_off[i] + in_len[i] < in_off[i] || out_off[i] + out_cap[i] < out_off[i]) return PZG_RC_BAD_ARG;
            if ((in_len[i] >> 40) || (out_cap[i] >> 40)) return PZG_RC_BAD_ARG;
            any_out |= out_cap[i] != 0;
        }
        if (any_out && !out_base) return PZG_RC_BAD_ARG;
        if (flags & PZG_HOST_PINNED) {  // the copy engines move whole spans of the arenas: extents ascend, none overlaps the next
            if (with_dict) return PZG_RC_BAD_ARG;
            for (uint32_t i = 0; i + 1 < n; ++i)
                if (in_off[i] + in_len[i] > in_off[i + 1] || out_off[i] + out_cap[i] > out_off[i + 1]) return PZG_RC_BAD_ARG;
        }
        HostBatch b{in_base, in_off, in_len, out_base, out_off, out_cap, out_len, status, detail, in_used, adler, flags};
        if (with_dict) {
            bool any = false;
            for (uint32_t i = 0; i < n; ++i) {
                if (dict_off[i] + dict_len[i] < dict_off[i] || (dict_len[i] >> 32)) return PZG_RC_BAD_ARG;
                any |= dict_len[i] != 0;
            }
            if (any) return dict_path(ctx, *ctx->shards[0], b, dict_base, dict_off, dict_len, n);
        }
        const size_t S = ctx->shards.size();
        if (S == 1) {
            std::vector<uint32_t> idx(n);
            std::iota(idx.begin(), idx.end(), 0u);
            if (!(flags & PZG_HOST_PINNED)) lpt_order(out_cap, idx);
            return host_path(ctx, *ctx->shards[0], b, idx.data(), n);
        }
        // several devices: longest-processing-time-first over the shards (by capacity), one host thread per shard
        std::vector<uint32_t> all(n);
        std::iota(all.begin(), all.end(), 0u);
        std::vector<std::vector<uint32_t>> part(S);
        if (flags & PZG_HOST_PINNED) {  // contiguous index slices of (nearly) equal load: every shard moves its own span of the arenas
            uint64_t total = 0, acc = 0;
            for (uint32_t i = 0; i < n; ++i) total += out_cap[i] + in_len[i] + 4096;
            size_t s = 0;
            for (uint32_t i = 0; i < n; ++i) {
                while (s + 1 < S && acc >= total * (s + 1) / S) ++s;
                part[s].push_back(i);
                acc += out_cap[i] + in_len[i] + 4096;
            }
        } else {
            lpt_order(out_cap, all);
            std::vector<uint64_t> load(S, 0);
            for (uint32_t i : all) {
                size_t best = 0;
                for (size_t s = 1; s < S; ++s)
                    if (load[s] < load[best]) best = s;
                part[best].push_back(i);
                load[best] += out_cap[i] + in_len[i] + 4096;  // (+ a per-stream constant: tiny streams are not free)
            }
        }
        std::vector<int> rcs(S, PZG_RC_OK);
        auto run_shard = [&](size_t s) noexcept {
            try {
                rcs[s] = host_path(ctx, *ctx->shards[s], b, part[s].data(), (uint32_t)part[s].size());
            } catch (...) {
                rcs[s] = PZG_RC_NO_MEMORY;
            }
        };
        std::vector<std::thread> th;
        th.reserve(S - 1);  // (emplace_back below cannot reallocate: a throw can only come from the thread's creation)
        size_t started = 1;
        for (; started < S; ++started) {
            try {
                th.emplace_back(run_shard, started);
            } catch (...) {  // no more threads to be had: this thread runs the remaining shards itself, one after the other
                break;
            }
        }
        run_shard(0);
        for (size_t s = started; s < S; ++s) run_shard(s);
        for (auto &t : th) t.join();  // always: the started threads write into the caller's buffers
        for (int rc : rcs)
            if (rc != PZG_RC_OK) return rc;
        return PZG_RC_OK;
    } catch (const std::bad_alloc &) {
        return PZG_RC_NO_MEMORY;
    } catch (...) {
        {
            std::lock_guard<std::mutex> g(ctx->err_mu);
            ctx->last_error = "C++ exception inside pzg_decompress_many";
        }
        return PZG_RC_HIP_ERROR;
    }
}

int pzg_decompress_many_sharded(pzg_ctx *ctx, const pzg_device_batch *batches, uint32_t nbatches, uint32_t flags)
{
    if (!ctx_live(ctx) || (!batches && nbatches)) return PZG_RC_BAD_ARG;
    if (flags & ~(PZG_ASYNC | PZG_GZIP | PZG_LPT_ORDER | PZG_DEVICE_PTRS)) return PZG_RC_BAD_ARG;
    for (uint32_t b = 0; b < nbatches; ++b) {
        const pzg_device_batch &q = batches[b];
        if (q.shard >= ctx->shards.size()) return PZG_RC_BAD_ARG;
        if (q.n && (!q.in_base || !q.in_off || !q.in_len || !q.out_base || !q.out_off || !q.out_cap || !q.out_len || !q.status))
            return PZG_RC_BAD_ARG;
    }
    try {
        // every batch enqueued on its own device's stream first (launches are asynchronous: one host thread feeds them all) ...
        int rc = PZG_RC_OK;
        for (uint32_t b = 0; b < nbatches && rc == PZG_RC_OK; ++b) {
            const pzg_device_batch &q = batches[b];
            if (q.n == 0) continue;
            pzg::InflateArgs a{};
            a.in_base = q.in_base;
            a.in_off = q.in_off;
            a.in_len = q.in_len;
            a.out_base = q.out_base;
            a.out_off = q.out_off;
            a.out_cap = q.out_cap;
            a.out_len = q.out_len;
            a.status = q.status;
            a.detail = q.detail;
            a.in_used = q.in_used;
            a.adler = q.adler;
            a.n = q.n;
            rc = launch_device(ctx, *ctx->shards[q.shard], a, (flags & (PZG_GZIP | PZG_LPT_ORDER)) | PZG_DEVICE_PTRS | PZG_ASYNC);
        }
        // ... then waited for, unless the caller does that (on an error too: nothing enqueued may outlive a failed call)
        if (!(flags & PZG_ASYNC) || rc != PZG_RC_OK) {
            for (auto &s : ctx->shards) {
                std::lock_guard<std::mutex> g(s->mu);
                if (hipSetDevice(s->device) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess)
                    if (rc == PZG_RC_OK) rc = PZG_RC_HIP_ERROR;
            }
        }
        return rc;
    } catch (...) {
        return PZG_RC_NO_MEMORY;
    }
}

// ---- resumable decoders (decompressIncremental) ---------------------------------------------------------------------
}  // extern "C"

struct pzg_decoder {
    pzg_ctx *ctx = nullptr;  // holds one reference (ctx_unref in pzg_decoder_destroy): never dangles
    int device = 0;
    uint32_t n = 0;
    size_t stride = 0;
    uint8_t *d_state = nullptr;  // n x stride: ResumeState + LDS image per decoder
    uint32_t *d_counter = nullptr;
    Arena d_in, d_out, d_meta, d_dense, d_doff;
    Arena d_strip;  // the resume kernels' scratch (round 5: spans of strips inside a feed): a slice per stream-wave of a feed's launches
    Pinned h_in, h_out, h_meta, h_doff;  // page-locked staging (grow-only): the copies run at link speed and really are asynchronous
    hipStream_t stream = nullptr;
    // a large feed is cut into ranges of decoders that overlap their uploads, launches, downloads and host copies
    static constexpr int MAXR = 8;
    hipStream_t s_up = nullptr, s_res = nullptr, s_dat = nullptr;
    hipStream_t s_kr[MAXR] = {};  // a launch stream per range: a range's decoders fill a sixth of the chip, the ranges' kernels run side by side
    hipEvent_t ev_up[MAXR] = {}, ev_k[MAXR] = {}, ev_res[MAXR] = {}, ev_dat[MAXR] = {};
    bool pipe_ready = false;
    // what the last large feed spent where, in ms (pzg_decoder_last_feed_ms): the call | packing | waiting for the ranges' kernels | downloads | copy-out
    double last_feed[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    std::mutex mu;
};

namespace {
int decoder_pipe_prepare(pzg_ctx *ctx, pzg_decoder *d)
{
    if (d->pipe_ready) return PZG_RC_OK;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIP_TRY(ctx, hipStreamCreateWithFlags(&d->s_up, hipStreamNonBlocking));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&d->s_res, hipStreamNonBlocking));
    HIP_TRY(ctx, hipStreamCreateWithPriority(&d->s_dat, hipStreamNonBlocking, least));  // (downloads: lowest, launches: highest, uploads: normal)
    for (int c = 0; c < pzg_decoder::MAXR; ++c) {
        HIP_TRY(ctx, hipStreamCreateWithPriority(&d->s_kr[c], hipStreamNonBlocking, greatest));  // (launch streams: see lane_prepare)
        HIP_TRY(ctx, hipEventCreateWithFlags(&d->ev_up[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&d->ev_k[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&d->ev_res[c], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&d->ev_dat[c], hipEventDisableTiming));
    }
    d->pipe_ready = true;
    return PZG_RC_OK;
}
}  // namespace

extern "C" {

int pzg_decoder_last_feed_ms(pzg_decoder *dec, double out[5])
{
    if (!dec || !out) return PZG_RC_BAD_ARG;
    std::lock_guard<std::mutex> g(dec->mu);
    for (int k = 0; k < 5; ++k) out[k] = dec->last_feed[k];
    return PZG_RC_OK;
}

int pzg_decoder_create(pzg_ctx *ctx, uint32_t n, pzg_decoder **out)
{
    if (out) *out = nullptr;
    if (!ctx_live(ctx) || !out || n == 0 || ctx->shards.size() != 1) return PZG_RC_BAD_ARG;
    pzg_decoder *d = nullptr;
    try {
        Shard &sh = *ctx->shards[0];
        HIP_TRY(ctx, hipSetDevice(sh.device));
        d = new pzg_decoder();
        ctx->refs.fetch_add(1);
        d->ctx = ctx;
        d->device = sh.device;
        d->n = n;
        d->stride = pzg::resume_state_bytes();
        // every failure from here on leaves through pzg_decoder_destroy: nothing allocated so far is leaked
        int rc = PZG_RC_OK;
        hipError_t e;
        if (hipMalloc((void **)&d->d_state, d->stride * (size_t)n) != hipSuccess || hipMalloc((void **)&d->d_counter, 256) != hipSuccess)
            rc = PZG_RC_NO_MEMORY;
        else if ((e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)) != hipSuccess)
            rc = hip_fail(ctx, e, "hipStreamCreateWithFlags");
        else if ((e = hipMemsetAsync(d->d_state, 0, d->stride * (size_t)n, d->stream)) != hipSuccess)
            rc = hip_fail(ctx, e, "hipMemsetAsync");
        else if ((e = hipStreamSynchronize(d->stream)) != hipSuccess)
            rc = hip_fail(ctx, e, "hipStreamSynchronize");
        if (rc != PZG_RC_OK) {
            pzg_decoder_destroy(d);
            return rc;
        }
        *out = d;
        return PZG_RC_OK;
    } catch (...) {
        if (d) pzg_decoder_destroy(d);
        return PZG_RC_NO_MEMORY;
    }
}

void pzg_decoder_destroy(pzg_decoder *dec)
{
    if (!dec) return;
    (void)hipSetDevice(dec->device);
    for (hipStream_t st : {dec->stream, dec->s_up, dec->s_res, dec->s_dat})
        if (st) (void)hipStreamSynchronize(st);
    for (hipStream_t st : dec->s_kr)
        if (st) (void)hipStreamSynchronize(st);
    for (Arena *a : {&dec->d_in, &dec->d_out, &dec->d_meta, &dec->d_dense, &dec->d_doff, &dec->d_strip})
        if (a->p) (void)hipFree(a->p);
    for (Pinned *h : {&dec->h_in, &dec->h_out, &dec->h_meta, &dec->h_doff}) pinned_release(*h);
    if (dec->d_state) (void)hipFree(dec->d_state);
    if (dec->d_counter) (void)hipFree(dec->d_counter);
    for (int c = 0; c < pzg_decoder::MAXR; ++c)
        for (hipEvent_t e : {dec->ev_up[c], dec->ev_k[c], dec->ev_res[c], dec->ev_dat[c]})
            if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {dec->s_up, dec->s_res, dec->s_dat})
        if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : dec->s_kr)
        if (st) (void)hipStreamDestroy(st);
    if (dec->stream) (void)hipStreamDestroy(dec->stream);
    pzg_ctx *ctx = dec->ctx;
    delete dec;
    if (ctx) ctx_unref(ctx);  // the last reference frees a context whose handle pzg_shutdown has already dropped
}

int pzg_decoder_reset(pzg_decoder *dec, const uint32_t *idx, uint32_t m)
{
    if (!dec) return PZG_RC_BAD_ARG;
    pzg_ctx *ctx = dec->ctx;
    std::lock_guard<std::mutex> g(dec->mu);
    HIP_TRY(ctx, hipSetDevice(dec->device));
    if (!idx) {
        HIP_TRY(ctx, hipMemsetAsync(dec->d_state, 0, dec->stride * (size_t)dec->n, dec->stream));
    } else {
        for (uint32_t j = 0; j < m; ++j) {
            if (idx[j] >= dec->n) return PZG_RC_BAD_ARG;
            HIP_TRY(ctx, hipMemsetAsync(dec->d_state + dec->stride * (size_t)idx[j], 0, pzg::resume_scalar_bytes(), dec->stream));
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(dec->stream));
    return PZG_RC_OK;
}

int pzg_decoder_feed(pzg_decoder *dec, const uint32_t *idx, uint32_t m, const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                     const uint8_t *final_in, uint8_t *out_base, const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                     int32_t *state, uint32_t *detail, uint64_t *in_used, uint32_t *chunks, uint32_t *adler)
{
    if (!dec || !in_off || !in_len || !out_base || !out_off || !out_cap || !out_len || !state || !in_used || !chunks)
        return PZG_RC_BAD_ARG;
    pzg_ctx *ctx = dec->ctx;
    if (!idx) m = dec->n;
    if (m == 0) return PZG_RC_OK;
    try {
        for (uint32_t j = 0; j < m; ++j) {
            if (idx && idx[j] >= dec->n) return PZG_RC_BAD_ARG;
            if (out_cap[j] < 4096u || (out_cap[j] >> 40) || (in_len[j] >> 40)) return PZG_RC_BAD_ARG;
            if (in_off[j] + in_len[j] < in_off[j] || out_off[j] + out_cap[j] < out_off[j]) return PZG_RC_BAD_ARG;
            if (in_len[j] && !in_base) return PZG_RC_BAD_ARG;  // (in_base may be NULL when no decoder has input: an empty tail)
        }
        if (idx) {  // a decoder may appear once per call (each is continued by one wave)
            std::vector<uint32_t> seen(idx, idx + m);
            std::sort(seen.begin(), seen.end());
            if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return PZG_RC_BAD_ARG;
        }
        std::lock_guard<std::mutex> g(dec->mu);
        const int num_cus = ctx->shards[0]->num_cus;
        HIP_TRY(ctx, hipSetDevice(dec->device));
        (void)hipStreamSynchronize(dec->stream);  // (idle unless an earlier call left on an error: nothing of it may still use the staging)
        // Packed staging in page-locked memory.  One meta block, both ways:
        //   in_off | in_len | out_off | out_cap : u64[m]   (host -> device)
        //   out_len | in_used : u64[m]; state | adler | chunks : 32-bit [m]; detail u32[2m]   (device -> host)
        //   final u8[m]   (host -> device)
        const size_t meta_bytes = 72 * (size_t)m + 64;
        int rc;
        if ((rc = pinned_reserve(ctx, dec->h_meta, meta_bytes)) != PZG_RC_OK) return rc;
        uint64_t *ioff = (uint64_t *)dec->h_meta.p, *ilen = ioff + m, *ooff = ilen + m, *ocap = ooff + m;
        size_t ip = 0, op = 0;
        for (uint32_t j = 0; j < m; ++j) {
            ioff[j] = ip;
            ilen[j] = in_len[j];
            ip += pad16(in_len[j]) + 16;
            ooff[j] = op;
            ocap[j] = out_cap[j];
            op += pad16(out_cap[j]);
        }
        uint8_t *h_final = dec->h_meta.p + 68 * (size_t)m;
        if (final_in) memcpy(h_final, final_in, m);
        else memset(h_final, 0, m);
        if ((rc = pinned_reserve(ctx, dec->h_in, ip + 64)) != PZG_RC_OK) return rc;
        if ((rc = pinned_reserve(ctx, dec->h_out, op + 64)) != PZG_RC_OK) return rc;
        if ((rc = arena_reserve(ctx, dec->d_in, ip + 64)) != PZG_RC_OK) return rc;
        if ((rc = arena_reserve(ctx, dec->d_out, op + 64)) != PZG_RC_OK) return rc;
        if ((rc = arena_reserve(ctx, dec->d_meta, meta_bytes)) != PZG_RC_OK) return rc;
        pzg::ResumeArgs a{};
        uint8_t *dm = (uint8_t *)dec->d_meta.p;
        a.state_stride = dec->stride;
        a.in_base = (const uint8_t *)dec->d_in.p;
        a.out_base = (uint8_t *)dec->d_out.p;
        a.in_off = (const uint64_t *)dm;
        a.in_len = a.in_off + m;
        a.out_off = a.in_len + m;
        a.out_cap = a.out_off + m;
        a.out_len = (uint64_t *)(dm + 32 * (size_t)m);
        a.in_used = a.out_len + m;
        a.status = (int32_t *)(a.in_used + m);
        a.adler = (uint32_t *)(a.status + m);
        a.chunks = a.adler + m;
        a.detail = a.chunks + m;
        a.final_in = dm + 68 * (size_t)m;
        a.counter = dec->d_counter;
        // one launch per run of consecutive decoder numbers among positions [j0, j1) of the call
        uint64_t *dense_host = nullptr;  // (set by the pipelined path when the kernels write their results into host memory)
        // The kernels' scratch: one slice per stream-wave of this feed's launches -- at most the chip's residency, at most an arena's
        // share of PZG_OPT_SCRATCH_BYTES; a feed that gets none (or fewer slices than waves) decodes those waves by windows alone.
        // Launches that may run side by side (the ranges of a large feed) get disjoint runs of slices.
        uint32_t strip_slices = 0;
        {
            const size_t per_wave = pzg::resume_strip_wave_bytes();
            size_t want = (size_t)pzg::resume_launch_waves(num_cus, m) * per_wave;
            const uint64_t capb = ctx->scratch_cap.load();
            if (capb != 0) {
                const size_t share = (size_t)(capb / (uint64_t)(STRIP_SLOTS + LANES_PER_SHARD)) / per_wave * per_wave;
                if (want > share) want = share;
                if (dec->d_strip.cap > share + 255u) {
                    (void)hipFree(dec->d_strip.p);
                    dec->d_strip.p = nullptr;
                    dec->d_strip.cap = 0;
                }
            }
            if (want >= per_wave) {
                if (dec->d_strip.cap < want) {
                    if (dec->d_strip.p) (void)hipFree(dec->d_strip.p);
                    dec->d_strip.p = nullptr;
                    dec->d_strip.cap = 0;
                    void *q = nullptr;
                    if (hipMalloc(&q, (want + 255u) & ~(size_t)255u) == hipSuccess) {
                        dec->d_strip.p = q;
                        dec->d_strip.cap = (want + 255u) & ~(size_t)255u;
                    } else {
                        (void)hipGetLastError();  // (tolerated: windows alone)
                    }
                }
                if (dec->d_strip.p) strip_slices = (uint32_t)(want / per_wave);
            }
        }
        uint32_t strip_lo = 0, strip_n = strip_slices;  // the run of slices the next launch_runs() call may use
        auto launch_runs = [&](uint32_t j0, uint32_t jend, uint32_t *counter, hipStream_t st, uint64_t dense_region = 0, uint32_t *dense_cursor = nullptr) -> hipError_t {
            if (dense_cursor) {
                const hipError_t e = hipMemsetAsync(dense_cursor, 0, sizeof(uint32_t), st);
                if (e != hipSuccess) return e;
            }
            while (j0 < jend) {
                uint32_t j1 = j0 + 1;
                const uint32_t first = idx ? idx[j0] : j0;
                if (!idx) j1 = jend;
                else
                    while (j1 < jend && idx[j1] == first + (j1 - j0)) ++j1;
                pzg::ResumeArgs r = a;
                r.state_base = dec->d_state + dec->stride * (size_t)first;
                r.in_off += j0;
                r.in_len += j0;
                r.out_off += j0;
                r.out_cap += j0;
                r.out_len += j0;
                r.in_used += j0;
                r.status += j0;
                r.adler += j0;
                r.chunks += j0;
                r.detail += 2 * (size_t)j0;
                r.final_in += j0;
                r.n = j1 - j0;
                r.counter = counter;
                if (strip_n != 0) {  // (the runs of one call go to one stream, one after the other: they share their slices)
                    r.strip = (uint32_t *)((uint8_t *)dec->d_strip.p + (size_t)strip_lo * pzg::resume_strip_wave_bytes());
                    r.strip_waves = strip_n;
                }
                if (dense_cursor) {
                    r.dense = (uint8_t *)dec->d_dense.p;
                    r.dense_region = dense_region;
                    r.dense_cursor = dense_cursor;
                    r.dense_off = (dense_host ? dense_host : (uint64_t *)dec->d_doff.p) + j0;
                }
                const hipError_t e = pzg::launch_resume(r, num_cus, st);
                if (e != hipSuccess) return e;
                j0 = j1;
            }
            return hipSuccess;
        };
        // ---- a large feed: ranges of decoders flow through upload -> launch -> results -> download -> copy-out, the stages
        // of neighbouring ranges overlapping (round 4: with the small-ring kernel the launches no longer dominate a feed)
        if (m >= 512u && ip + op >= (64ull << 20)) {
            if ((rc = decoder_pipe_prepare(ctx, dec)) != PZG_RC_OK) return rc;
            // (what the decoders deliver is packed back to back on the device and comes down in one linear copy per range)
            if ((rc = arena_reserve(ctx, dec->d_dense, op + 64)) != PZG_RC_OK) return rc;
            if ((rc = arena_reserve(ctx, dec->d_doff, 8 * (size_t)m + 64)) != PZG_RC_OK) return rc;
            if ((rc = pinned_reserve(ctx, dec->h_doff, 8 * (size_t)m + 64)) != PZG_RC_OK) return rc;
            uint64_t *doff = (uint64_t *)dec->h_doff.p;  // where decoder j's bytes start in the packed buffers (device and host alike)
            constexpr uint32_t R = 8;
            static_assert(R <= (uint32_t)pzg_decoder::MAXR, "an event set per range");
            // (what does not overlap is the first range's upload + decode and the last range's copy-out: the ranges shrink towards the end)
            static const uint32_t WEIGHT[R] = {10, 10, 9, 8, 7, 6, 5, 4};
            uint32_t lo[R + 1];
            {
                uint32_t wsum = 0, wacc[R];
                for (uint32_t c = 0; c < R; ++c) wacc[c] = (wsum += WEIGHT[c]);
                const size_t tot = ip + op;
                size_t acc = 0;
                uint32_t c = 0;
                lo[0] = 0;
                for (uint32_t j = 0; j < m; ++j) {
                    acc += pad16(in_len[j]) + 16 + pad16(out_cap[j]);
                    while (c + 1 < R && acc >= tot / wsum * wacc[c]) lo[++c] = j + 1;
                }
                while (c < R) lo[++c] = m;
            }
            uint8_t *hin = dec->h_in.p, *hout = dec->h_out.p;
            uint8_t *res = dec->h_meta.p + 32 * (size_t)m;
            uint64_t *olen = (uint64_t *)res, *used = olen + m;
            int32_t *stt = (int32_t *)(used + m);
            uint32_t *ad = (uint32_t *)(stt + m), *ch = ad + m, *det = ch + m;
            // The results (44 bytes per decoder) are written by the kernels STRAIGHT into the page-locked meta block -- it is mapped
            // into the device's address space -- so that they are there when a range's kernel ends: as copies they queued up
            // behind the megabytes of the ranges in front on the one download engine.  (Pageable fallback staging: copies.)
            const bool direct = !dec->h_meta.pageable && !dec->h_doff.pageable;
            if (direct) {
                dense_host = doff;
                a.out_len = olen;
                a.in_used = used;
                a.status = stt;
                a.adler = ad;
                a.chunks = ch;
                a.detail = det;
            }
            hipError_t herr = hipSuccess;
            const char *hwhat = "";
#define FEED_TRY(call)                              \
    do {                                            \
        if (herr == hipSuccess) {                   \
            herr = (call);                          \
            if (herr != hipSuccess) hwhat = #call;  \
        }                                           \
    } while (0)
            FEED_TRY(hipMemcpyAsync(dm, dec->h_meta.p, 32 * (size_t)m, hipMemcpyHostToDevice, dec->s_up));
            FEED_TRY(hipMemcpyAsync(dm + 68 * (size_t)m, h_final, m, hipMemcpyHostToDevice, dec->s_up));
            std::mutex pm;
            std::condition_variable pcv;
            uint32_t issued = 0, fetched = 0;
            bool stop = false, stop_copy = false;
            hipError_t derr = hipSuccess;
#if defined(PZG_LAB)
            const bool trace = getenv("PZG_TRACE_HOST") != nullptr;
#endif
            // (a handful of clock reads per range: what pzg_decoder_last_feed_ms reports)
            const auto t_feed0 = std::chrono::steady_clock::now();
            auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
            double t_pack = 0, t_res = 0, t_dat = 0, t_out = 0;
            // one fetched range handed to the caller (results into the caller's arrays, bytes out of the staging)
            auto copy_one = [&](uint32_t c) {
                const uint32_t j0 = lo[c], j1 = lo[c + 1];
                if (j0 == j1) return;
                const auto t0 = std::chrono::steady_clock::now();
                uint64_t delivered = 0;
                for (uint32_t j = j0; j < j1; ++j) {
                    out_len[j] = olen[j];
                    state[j] = stt[j];
                    in_used[j] = used[j];
                    chunks[j] = ch[j];
                    if (adler) adler[j] = ad[j];
                    if (detail) {
                        detail[2 * (size_t)j] = det[2 * (size_t)j];
                        detail[2 * (size_t)j + 1] = det[2 * (size_t)j + 1];
                    }
                    delivered += olen[j] <= ocap[j] ? olen[j] : ocap[j];
                }
                const uint32_t nn = j1 - j0;
                ctx->helpers->run(delivered >= (4u << 20) ? (ctx->helpers->size() + 1u) / 2u : 1u, [&](unsigned part, unsigned nparts) {
                    for (uint32_t j = j0 + (uint32_t)((uint64_t)nn * part / nparts), e2 = j0 + (uint32_t)((uint64_t)nn * (part + 1) / nparts); j < e2; ++j)
                        if (olen[j]) memcpy(out_base + out_off[j], hout + doff[j], olen[j] <= ocap[j] ? olen[j] : ocap[j]);
                });
                t_out += ms_since(t0);
            };
            // one issued range fetched: wait for its results, bring down what its decoders delivered
            auto fetch_one = [&](uint32_t c) -> hipError_t {
                const uint32_t j0 = lo[c], j1 = lo[c + 1];
                if (j0 == j1) return hipSuccess;
                auto t0 = std::chrono::steady_clock::now();
                hipError_t e = hipEventSynchronize(dec->ev_res[c]);
                t_res += ms_since(t0);
                t0 = std::chrono::steady_clock::now();
                uint64_t delivered = 0;  // the decoders of the range packed their bytes behind one another from ooff[j0] on (whole 16-byte vectors each)
                for (uint32_t j = j0; j < j1 && e == hipSuccess; ++j) delivered += pad16(olen[j]);
                if (e == hipSuccess && delivered != 0) {
                    e = hipMemcpyAsync(hout + ooff[j0], (const uint8_t *)dec->d_dense.p + ooff[j0], delivered, hipMemcpyDeviceToHost, dec->s_dat);
                    if (e == hipSuccess) e = hipStreamSynchronize(dec->s_dat);
                }
                t_dat += ms_since(t0);
                return e;
            };
            // The second and third thread of the call (joined whatever way the call ends; an exception inside one -- the helpers'
            // job list can run out of memory -- is reported through derr, never thrown across the thread boundary).  Without them
            // (the system has no thread to give) the issuing thread fetches and copies the ranges itself once all are issued.
            CallThread copier, drainer;
            copier.wake = [&] {
                {
                    std::lock_guard<std::mutex> g(pm);
                    stop_copy = true;
                }
                pcv.notify_all();
            };
            drainer.wake = [&] {
                {
                    std::lock_guard<std::mutex> g(pm);
                    stop = true;
                }
                pcv.notify_all();
            };
            auto copier_body = [&] {
                for (uint32_t c = 0; c < R; ++c) {
                    {
                        std::unique_lock<std::mutex> g(pm);
                        pcv.wait(g, [&] { return fetched > c || stop_copy; });
                        if (fetched <= c) return;
                    }
                    try {
                        copy_one(c);
                    } catch (...) {
                        std::lock_guard<std::mutex> g(pm);
                        if (derr == hipSuccess) derr = hipErrorOutOfMemory;
                        return;
                    }
                }
            };
            auto drainer_body = [&] {
                (void)hipSetDevice(dec->device);
                for (uint32_t c = 0; c < R; ++c) {
                    {
                        std::unique_lock<std::mutex> g(pm);
                        pcv.wait(g, [&] { return issued > c || stop; });
                        if (issued <= c) return;
                    }
                    const hipError_t e = fetch_one(c);
                    {
                        std::lock_guard<std::mutex> g(pm);
                        if (e != hipSuccess && derr == hipSuccess) derr = e;
                        if (e == hipSuccess) fetched = c + 1;
                        else stop_copy = true;
                    }
                    pcv.notify_all();
                    if (e != hipSuccess) return;
                }
            };
            bool threaded = copier.start(copier_body);
            if (threaded && !drainer.start(drainer_body)) {
                copier.wake();
                copier.join();
                threaded = false;
            }
            for (uint32_t c = 0; c < R && herr == hipSuccess; ++c) {
                const uint32_t j0 = lo[c], j1 = lo[c + 1];
                if (j0 != j1) {
                    const uint32_t nn = j1 - j0;
                    const size_t ib = (j1 < m ? ioff[j1] : ip) - ioff[j0];
                    const auto tp0 = std::chrono::steady_clock::now();
                    ctx->helpers->run(ib >= (4u << 20) ? ctx->helpers->size() : 1u, [&](unsigned part, unsigned nparts) {
                        for (uint32_t j = j0 + (uint32_t)((uint64_t)nn * part / nparts), e2 = j0 + (uint32_t)((uint64_t)nn * (part + 1) / nparts); j < e2; ++j)
                            if (ilen[j]) memcpy(hin + ioff[j], in_base + in_off[j], ilen[j]);
                    });
                    t_pack += ms_since(tp0);
                    if (ib) FEED_TRY(hipMemcpyAsync((uint8_t *)dec->d_in.p + ioff[j0], hin + ioff[j0], ib, hipMemcpyHostToDevice, dec->s_up));
                    FEED_TRY(hipEventRecord(dec->ev_up[c], dec->s_up));
                    hipStream_t s_k = dec->s_kr[c];
                    FEED_TRY(hipStreamWaitEvent(s_k, dec->ev_up[c], 0));
                    // (a counter word, a packing cursor and a stream of its own per range: the launches overlap; the range's packed region
                    // starts where its rooms start)
                    strip_lo = (uint32_t)((uint64_t)strip_slices * j0 / m);  // (ranges run side by side: each its own slices)
                    strip_n = (uint32_t)((uint64_t)strip_slices * j1 / m) - strip_lo;
                    FEED_TRY(launch_runs(j0, j1, dec->d_counter + 8u * c, s_k, ooff[j0], dec->d_counter + 8u * c + 1u));
                    FEED_TRY(hipEventRecord(dec->ev_k[c], s_k));
                    if (direct) {
                        FEED_TRY(hipEventRecord(dec->ev_res[c], s_k));
                    } else {
                    FEED_TRY(hipStreamWaitEvent(dec->s_res, dec->ev_k[c], 0));
                    // the range's results: six short arrays (36 bytes per decoder)
                    FEED_TRY(hipMemcpyAsync(olen + j0, a.out_len + j0, 8 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipMemcpyAsync(used + j0, a.in_used + j0, 8 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipMemcpyAsync(stt + j0, a.status + j0, 4 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipMemcpyAsync(ad + j0, a.adler + j0, 4 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipMemcpyAsync(ch + j0, a.chunks + j0, 4 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipMemcpyAsync(det + 2 * (size_t)j0, a.detail + 2 * (size_t)j0, 8 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipMemcpyAsync(doff + j0, (const uint64_t *)dec->d_doff.p + j0, 8 * (size_t)nn, hipMemcpyDeviceToHost, dec->s_res));
                    FEED_TRY(hipEventRecord(dec->ev_res[c], dec->s_res));
                    }
                }
                if (herr == hipSuccess) {
                    {
                        std::lock_guard<std::mutex> g(pm);
                        issued = c + 1;
                    }
                    pcv.notify_all();
                }
            }
#undef FEED_TRY
            if (threaded) {
                drainer.wake();
                drainer.join();
                copier.wake();  // (a no-op when every range was fetched: the copier finishes them all first)
                copier.join();
            } else if (herr == hipSuccess) {
                for (uint32_t c = 0; c < R && derr == hipSuccess; ++c) {
                    derr = fetch_one(c);
                    if (derr == hipSuccess) copy_one(c);  // (an exception here is the issuing thread's: the call's own catch reports it)
                }
            }
            dec->last_feed[0] = ms_since(t_feed0);
            dec->last_feed[1] = t_pack;
            dec->last_feed[2] = t_res;
            dec->last_feed[3] = t_dat;
            dec->last_feed[4] = t_out;
#if defined(PZG_LAB)
            if (trace)
                fprintf(stderr, "[pzg] feed: %u decoders, %.1f MiB in, %.1f MiB of rooms: %.1f ms (issuing thread: packing %.1f; draining thread: waiting for "
                        "results %.1f, fetching data %.1f, copy-out %.1f)\n", m, ip / 1048576.0, op / 1048576.0, ms_since(t_feed0), t_pack, t_res, t_dat, t_out);
#endif
            if (herr == hipSuccess && derr != hipSuccess) {
                herr = derr;
                hwhat = "download of a range";
            }
            if (herr != hipSuccess) {
                for (hipStream_t st : {dec->s_up, dec->s_res, dec->s_dat}) (void)hipStreamSynchronize(st);
                for (hipStream_t st : dec->s_kr) (void)hipStreamSynchronize(st);
                return hip_fail(ctx, herr, hwhat);
            }
            return PZG_RC_OK;
        }
        // ---- a small feed: one stream, one range
        // the decoders' inputs, packed by the context's helper threads (a batch of thousands of decoders moves 100+ MiB)
        const unsigned parts = ip >= (8u << 20) ? ctx->helpers->size() : 1u;
        uint8_t *hin = dec->h_in.p;
        ctx->helpers->run(parts, [&](unsigned part, unsigned nparts) {
            for (uint32_t j = (uint32_t)((uint64_t)m * part / nparts), e = (uint32_t)((uint64_t)m * (part + 1) / nparts); j < e; ++j)
                if (ilen[j]) memcpy(hin + ioff[j], in_base + in_off[j], ilen[j]);
        });
        hipStream_t st = dec->stream;
        HIP_TRY(ctx, hipMemcpyAsync(dm, dec->h_meta.p, 32 * (size_t)m, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(dm + 68 * (size_t)m, h_final, m, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(dec->d_in.p, hin, ip, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, launch_runs(0, m, dec->d_counter, st));
        // results first (36 bytes per decoder), then only what the decoders delivered: a decoder's room is out_cap, what it
        // fills of it is usually far less -- one copy over the whole room when most of it is used, else one per decoder run
        uint8_t *res = dec->h_meta.p + 32 * (size_t)m;
        HIP_TRY(ctx, hipMemcpyAsync(res, dm + 32 * (size_t)m, 36 * (size_t)m, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        const uint64_t *olen = (const uint64_t *)res, *used = olen + m;
        const int32_t *stt = (const int32_t *)(used + m);
        const uint32_t *ad = (const uint32_t *)(stt + m), *ch = ad + m, *det = ch + m;
        uint8_t *hout = dec->h_out.p;
        uint64_t delivered = 0;
        for (uint32_t j = 0; j < m; ++j) delivered += olen[j] <= ocap[j] ? olen[j] : ocap[j];
        bool same_room = true;
        uint64_t most = 0;
        for (uint32_t j = 0; j < m; ++j) {
            same_room = same_room && ocap[j] == ocap[0];
            const uint64_t dj = olen[j] <= ocap[j] ? olen[j] : ocap[j];
            most = dj > most ? dj : most;
        }
        if (delivered == 0) {
            // nothing to fetch
        } else if (delivered * 4 >= op * 3 || (!same_room && m > 64u)) {
            HIP_TRY(ctx, hipMemcpyAsync(hout, dec->d_out.p, op, hipMemcpyDeviceToHost, st));
        } else if (same_room) {  // equal rooms (the mirrors' case): one strided copy of the used part of every room
            HIP_TRY(ctx, hipMemcpy2DAsync(hout, pad16(ocap[0]), dec->d_out.p, pad16(ocap[0]), pad16(most), m, hipMemcpyDeviceToHost, st));
        } else {
            for (uint32_t j = 0; j < m; ++j)
                if (olen[j]) HIP_TRY(ctx, hipMemcpyAsync(hout + ooff[j], (const uint8_t *)dec->d_out.p + ooff[j], pad16(olen[j] <= ocap[j] ? olen[j] : ocap[j]), hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
        for (uint32_t j = 0; j < m; ++j) {
            out_len[j] = olen[j];
            state[j] = stt[j];
            in_used[j] = used[j];
            chunks[j] = ch[j];
            if (adler) adler[j] = ad[j];
            if (detail) {
                detail[2 * (size_t)j] = det[2 * (size_t)j];
                detail[2 * (size_t)j + 1] = det[2 * (size_t)j + 1];
            }
        }
        ctx->helpers->run(delivered >= (8u << 20) ? ctx->helpers->size() : 1u, [&](unsigned part, unsigned nparts) {
            for (uint32_t j = (uint32_t)((uint64_t)m * part / nparts), e = (uint32_t)((uint64_t)m * (part + 1) / nparts); j < e; ++j)
                if (olen[j]) memcpy(out_base + out_off[j], hout + ooff[j], olen[j] <= ocap[j] ? olen[j] : ocap[j]);
        });
        return PZG_RC_OK;
    } catch (const std::bad_alloc &) {
        return PZG_RC_NO_MEMORY;
    } catch (...) {
        return PZG_RC_HIP_ERROR;
    }
}

int pzg_decompress(pzg_ctx *ctx, const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                   uint64_t *out_len, int32_t *status, uint32_t detail[2], uint64_t *in_used)
{
    if (!out_len || !status) return PZG_RC_BAD_ARG;
    static const uint8_t dummy_in = 0;
    const uint64_t zero = 0;
    if (!out) out_cap = 0;  // count only: nothing is stored
    if (!in) in_len = 0;
    return pzg_decompress_many(ctx, in ? in : &dummy_in, &zero, &in_len, out, &zero, &out_cap, out_len, status, detail, in_used,
                               nullptr, 1, 0);
}

int pzg_adler32(pzg_ctx *ctx, const uint8_t *buf, uint64_t len, uint32_t init, uint32_t *out, uint32_t flags)
{
    if (!ctx_live(ctx) || !out || (!buf && len)) return PZG_RC_BAD_ARG;
    if ((flags & PZG_ASYNC) && !(flags & PZG_DEVICE_PTRS)) return PZG_RC_BAD_ARG;
    try {
        Shard &sh = *ctx->shards[0];
        std::lock_guard<std::mutex> g(sh.mu);
        HIP_TRY(ctx, hipSetDevice(sh.device));
        int rc;
        if ((rc = arena_reserve(ctx, sh.a_adler, 12 * (size_t)(ADLER_MAX_WAVES + 4) + 16)) != PZG_RC_OK) return rc;
        uint32_t *partials = (uint32_t *)sh.a_adler.p;
        uint32_t *d_res = partials + 3 * (size_t)(ADLER_MAX_WAVES + 4);
        hipStream_t s = sh.stream;
        if (flags & PZG_DEVICE_PTRS) {
            HIP_TRY(ctx, hipEventRecord(sh.ev0, s));
            HIP_TRY(ctx, pzg::launch_adler32(buf, len, init, partials, ADLER_MAX_WAVES, out, s));
            HIP_TRY(ctx, hipEventRecord(sh.ev1, s));
            sh.timed = true;
            sh.last_was_host = false;
            if (!(flags & PZG_ASYNC)) HIP_TRY(ctx, hipStreamSynchronize(s));
            return PZG_RC_OK;
        }
        const uint32_t skew = (uint32_t)((uintptr_t)buf & 15u);
        if ((rc = arena_reserve(ctx, sh.a_scratch, len + 64)) != PZG_RC_OK) return rc;
        uint8_t *d_buf = (uint8_t *)sh.a_scratch.p + skew;
        if (len) HIP_TRY(ctx, hipMemcpyAsync(d_buf, buf, len, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipEventRecord(sh.ev0, s));
        HIP_TRY(ctx, pzg::launch_adler32(d_buf, len, init, partials, ADLER_MAX_WAVES, d_res, s));
        HIP_TRY(ctx, hipEventRecord(sh.ev1, s));
        sh.timed = true;
        sh.last_was_host = false;
        HIP_TRY(ctx, hipMemcpyAsync(out, d_res, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        return PZG_RC_OK;
    } catch (...) {
        return PZG_RC_NO_MEMORY;
    }
}

int pzg_adler32_many(pzg_ctx *ctx, const uint8_t *base, const uint64_t *off, const uint64_t *len, uint32_t *out, uint32_t n,
                     uint32_t flags)
{
    if (!ctx_live(ctx) || !base || !off || !len || !out) return PZG_RC_BAD_ARG;
    if (!(flags & PZG_DEVICE_PTRS) || ctx->shards.size() != 1) return PZG_RC_BAD_ARG;  // device-resident batches only
    if (n == 0) return PZG_RC_OK;
    Shard &sh = *ctx->shards[0];
    std::lock_guard<std::mutex> g(sh.mu);
    HIP_TRY(ctx, hipSetDevice(sh.device));
    HIP_TRY(ctx, hipEventRecord(sh.ev0, sh.stream));
    HIP_TRY(ctx, pzg::launch_adler32_many(base, off, len, out, n, sh.num_cus, sh.stream));
    HIP_TRY(ctx, hipEventRecord(sh.ev1, sh.stream));
    sh.timed = true;
    sh.last_was_host = false;
    if (!(flags & PZG_ASYNC)) HIP_TRY(ctx, hipStreamSynchronize(sh.stream));
    return PZG_RC_OK;
}

double pzg_last_kernel_ms(pzg_ctx *ctx)
{
    if (!ctx_live(ctx)) return -1.0;
    Shard &sh = *ctx->shards[0];
    std::lock_guard<std::mutex> g(sh.mu);
    if (!sh.timed) return -1.0;
    if (sh.last_was_host) return sh.host_ms;
    if (hipSetDevice(sh.device) != hipSuccess) return -1.0;
    if (hipEventSynchronize(sh.ev1) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, sh.ev0, sh.ev1) != hipSuccess) return -1.0;
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

const char *pzg_last_error(pzg_ctx *ctx)
{
    if (!ctx) return "";
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(ctx->err_mu);
    copy = ctx->last_error;
    return copy.c_str();
}

uint32_t pzg_version(void) { return (PZG_VERSION_MAJOR << 16) | PZG_VERSION_MINOR; }

}  // extern "C"
