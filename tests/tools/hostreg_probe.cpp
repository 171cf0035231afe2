// Diagnostic: what would downloading straight into the caller's arena cost?  A 2 GiB pageable host
// buffer: hipHostRegister / hipHostUnregister time, D2H into it once registered, against D2H into page-locked staging
// (hipHostMalloc) plus the memcpy out of it (one thread / 8 threads), and plain pageable hipMemcpy.
//   hipcc -O2 --offload-arch=gfx950 tests/tools/hostreg_probe.cpp -o build/probe/hostreg_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t N = 2ull << 30;
    void *d = nullptr;
    CK(hipMalloc(&d, N));
    CK(hipMemset(d, 0x5a, N));
    uint8_t *pageable = (uint8_t *)malloc(N);
    memset(pageable, 1, N);  // touched: the pages exist
    uint8_t *pinned = nullptr;
    CK(hipHostMalloc((void **)&pinned, 256u << 20, hipHostMallocDefault));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipHostRegister(pageable, N, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(pageable, d, N, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(pageable));
        double t3 = now();
        printf("rep %d: register %.1f ms, D2H into registered %.1f ms (%.1f GB/s), unregister %.1f ms, total %.1f ms\n", rep, t1 - t0, t2 - t1,
               N / (t2 - t1) / 1e6, t3 - t2, t3 - t0);
    }
    for (int threads : {1, 8}) {  // staged: 8 x 256 MiB through one pinned slot (no overlap), then with the copy-out overlapped
        double t0 = now();
        for (size_t off = 0; off < N; off += 256u << 20) {
            CK(hipMemcpyAsync(pinned, (uint8_t *)d + off, 256u << 20, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            std::vector<std::thread> th;
            const size_t part = (256u << 20) / threads;
            for (int t = 0; t < threads; ++t) th.emplace_back([=] { memcpy(pageable + off + t * part, pinned + t * part, part); });
            for (auto &x : th) x.join();
        }
        double t1 = now();
        printf("staged through 256 MiB pinned, %d copy thread(s), no overlap: %.1f ms (%.1f GB/s)\n", threads, t1 - t0, N / (t1 - t0) / 1e6);
    }
    {
        double t0 = now();
        CK(hipMemcpy(pageable, d, N, hipMemcpyDeviceToHost));
        double t1 = now();
        printf("plain hipMemcpy into pageable: %.1f ms (%.1f GB/s)\n", t1 - t0, N / (t1 - t0) / 1e6);
    }
    {
        uint8_t *big = nullptr;
        double t0 = now();
        CK(hipHostMalloc((void **)&big, N, hipHostMallocDefault));
        double t1 = now();
        CK(hipMemcpyAsync(big, d, N, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        printf("hipHostMalloc 2 GiB %.1f ms; D2H into it %.1f ms (%.1f GB/s)\n", t1 - t0, t2 - t1, N / (t2 - t1) / 1e6);
    }
    return 0;
}
