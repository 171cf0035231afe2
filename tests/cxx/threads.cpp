// threads.cpp -- several host threads share one pzg_ctx (the library locks it internally: SURVEY.md 8b "Threading") and
// decode the reference fixtures concurrently; every result must be right.  Usage: threads <name.z> <name.gold> ...
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pzg.h"

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(2);
    }
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    pzg_ctx *ctx = nullptr;
    if (pzg_init(0, &ctx) != PZG_RC_OK) return 3;
    std::vector<std::vector<uint8_t>> zs, golds;
    for (int i = 1; i + 1 < argc; i += 2) {
        zs.push_back(slurp(argv[i]));
        golds.push_back(slurp(argv[i + 1]));
    }
    std::atomic<int> bad{0}, done{0};
    auto worker = [&](int t, int reps) {
        for (int rep = 0; rep < reps; ++rep) {
            const size_t k = (size_t)(t * 7 + rep) % zs.size();
            std::vector<uint8_t> out(golds[k].size() + 64);
            uint64_t out_len = 0, in_used = 0;
            int32_t status = -1;
            uint32_t detail[2] = {0, 0};
            const int rc = pzg_decompress(ctx, zs[k].data(), zs[k].size(), out.data(), golds[k].size(), &out_len, &status, detail, &in_used);
            if (rc != PZG_RC_OK || status != PZG_OK || out_len != golds[k].size() || memcmp(out.data(), golds[k].data(), golds[k].size()) != 0)
                ++bad;
            ++done;
        }
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    {  // warm every pipeline of the context (streams, arenas, staging, token scratch): eight calls at once take eight of them
        std::vector<std::thread> warm;
        for (int t = 0; t < 8; ++t) warm.emplace_back(worker, t, 3);
        for (auto &th : warm) th.join();
    }
    done = 0;
    // the same 160 calls from one thread, then from eight
    const auto t0 = now();
    for (int t = 0; t < 8; ++t) worker(t, 20);
    const double serial = std::chrono::duration<double>(now() - t0).count();
    done = 0;
    const auto t1 = now();
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t) pool.emplace_back(worker, t, 20);
    for (auto &th : pool) th.join();
    const double threaded = std::chrono::duration<double>(now() - t1).count();
    pzg_shutdown(ctx);
    printf("threads: %d calls from 8 threads, %d bad\n", done.load(), bad.load());
    printf("one thread %.1f ms, eight threads %.1f ms, speed-up over one thread: %.2fx\n", serial * 1e3, threaded * 1e3, serial / threaded);
    return bad ? 1 : 0;
}
