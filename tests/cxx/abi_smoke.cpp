// abi_smoke.cpp -- exercises the C ABI of include/pzg.h from plain C++ (no Python, no torch):
// decodes every reference fixture given on the command line as <name.z> <name.gold> pairs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    int rc = pzg_init(0, &ctx);
    if (rc != PZG_RC_OK) {
        fprintf(stderr, "pzg_init: %s\n", pzg_strerror(rc));
        return 3;
    }
    int bad = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::vector<uint8_t> z = slurp(argv[i]), gold = slurp(argv[i + 1]);
        std::vector<uint8_t> out(gold.size() + 64);
        uint64_t out_len = 0, in_used = 0;
        int32_t status = -1;
        uint32_t detail[2] = {0, 0};
        rc = pzg_decompress(ctx, z.data(), z.size(), out.data(), gold.size(), &out_len, &status, detail, &in_used);
        bool ok = rc == PZG_RC_OK && status == PZG_OK && out_len == gold.size() &&
                  memcmp(out.data(), gold.data(), gold.size()) == 0 && in_used == z.size();
        char msg[256];
        pzg_error_message(z.data(), z.size(), status, detail, msg, sizeof msg);
        printf("%-40s rc=%d status=%d out_len=%llu in_used=%llu %s %s  kernel %.3f ms\n", argv[i], rc, status,
               (unsigned long long)out_len, (unsigned long long)in_used, ok ? "OK" : "MISMATCH", msg,
               pzg_last_kernel_ms(ctx));
        if (!ok) bad++;
    }
    pzg_shutdown(ctx);
    return bad ? 1 : 0;
}
