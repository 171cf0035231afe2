// sanitize_main.cpp -- TEST INFRASTRUCTURE: runs the oracle and the kernel host model (inflate_core.h
// compiled for the host) over the streams named on the command line under ASan + UBSan.
// GPU AddressSanitizer is not available on this pool; the kernel's indexing logic is the same source.
//   usage: sanitize_main <capacity> <ring_bits> file.z [file.z ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/pz_oracle.h"

struct pzm_result {
    int32_t status;
    uint32_t detail0, detail1, adler;
    uint64_t out_len, in_used;
};
extern "C" int pzm_decompress(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t cap, int ring_bits, pzm_result *r);

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const uint64_t cap = strtoull(argv[1], nullptr, 10);
    const int rb = atoi(argv[2]);
    int bad = 0;
    for (int i = 3; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) return 2;
        std::vector<uint8_t> z;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) z.insert(z.end(), buf, buf + n);
        fclose(f);
        std::vector<uint8_t> o1(cap ? cap : 1), o2(cap ? cap : 1);
        pzo_result ro;
        pzm_result rm;
        pzo_decompress(z.data(), z.size(), o1.data(), cap, &ro);
        pzm_decompress(z.data(), z.size(), o2.data(), cap, rb, &rm);
        const uint64_t m = ro.out_len < cap ? ro.out_len : cap;
        if (ro.status != rm.status || (ro.status == 0 && (ro.out_len != rm.out_len || memcmp(o1.data(), o2.data(), m) != 0))) {
            printf("MISMATCH %s: oracle %d model %d\n", argv[i], ro.status, rm.status);
            bad++;
        }
    }
    printf("checked %d streams, %d mismatches\n", argc - 3, bad);
    return bad ? 1 : 0;
}
