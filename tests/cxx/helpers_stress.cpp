// helpers_stress.cpp -- TEST INFRASTRUCTURE (CPU, ThreadSanitizer): the helper thread pool of the host-pointer paths
// (pure_zlib_amd/csrc/pzg_helpers.h) hammered the way pzg_api.cpp uses it -- several threads calling run() at once with
// jobs of different sizes, every part exactly once, results visible to the caller when run() returns, clean shutdown with
// jobs in flight right up to it.
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "../../pure_zlib_amd/csrc/pzg_helpers.h"

int main(int argc, char **argv)
{
    const unsigned workers = argc > 1 ? (unsigned)atoi(argv[1]) : 8u, rounds = argc > 2 ? (unsigned)atoi(argv[2]) : 300u;
    long bad = 0;
    {
        Helpers pool(workers);
        std::vector<std::thread> callers;
        std::atomic<long> errors{0};
        for (unsigned c = 0; c < 4; ++c)
            callers.emplace_back([&, c] {
                for (unsigned r = 0; r < rounds; ++r) {
                    const unsigned parts = 1u + (r * 7u + c * 3u) % 37u;
                    std::vector<unsigned> hits(parts, 0u);          // written by whoever runs the part, read by the caller afterwards
                    std::vector<unsigned long> sums(parts, 0ul);
                    pool.run(parts, [&](unsigned p, unsigned n) {
                        if (n != parts || p >= parts) errors++;
                        hits[p] += 1u;
                        unsigned long s = 0;
                        for (unsigned k = 0; k < 2000u + 50u * p; ++k) s += k * (p + 1u);
                        sums[p] = s;
                    });
                    for (unsigned p = 0; p < parts; ++p) {
                        unsigned long s = 0;
                        for (unsigned k = 0; k < 2000u + 50u * p; ++k) s += k * (p + 1u);
                        if (hits[p] != 1u || sums[p] != s) errors++;
                    }
                }
            });
        for (auto &t : callers) t.join();
        bad = errors.load();
        pool.run(64, [](unsigned, unsigned) {});  // (the pool goes out of scope right behind a job)
    }
    Helpers one(1);  // a pool of one: the caller does everything itself
    unsigned n1 = 0;
    one.run(5, [&](unsigned, unsigned) { n1++; });
    if (n1 != 5) bad++;
    printf("helpers stress: %ld error(s)\n", bad);
    return bad ? 1 : 0;
}
