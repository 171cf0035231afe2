// lifetimes.cpp -- object lifetimes at the C ABI (include/pzg.h "Lifetimes").  The reference's ZlibDecoder is a
// garbage-collected closure (src/Codec/Compression/Zlib/Monad.hs:163-197): it can be dropped at any time and in any
// order.  Here: decoders and the context they came from are destroyed in both orders, a decoder is USED after
// pzg_shutdown, and the invalidated handle is refused instead of dereferenced.
//   lifetimes <name.z> <name.gold>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) {                                                \
            fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); \
            return 1;                                              \
        }                                                          \
    } while (0)

// feeds the whole stream through decoder 0 of `dec` in `piece`-byte feeds; returns 0 when the output equals gold
static int run_stream(pzg_decoder *dec, const std::vector<uint8_t> &z, const std::vector<uint8_t> &gold, size_t piece)
{
    std::vector<uint8_t> got, tail;
    size_t fed = 0;
    int32_t st = PZG_DEC_NEED_INPUT;
    while (st == PZG_DEC_NEED_INPUT || st == PZG_DEC_OUT_FULL) {
        if (st == PZG_DEC_NEED_INPUT) {
            CHECK(fed < z.size());
            const size_t k = z.size() - fed < piece ? z.size() - fed : piece;
            tail.insert(tail.end(), z.begin() + fed, z.begin() + fed + k);
            fed += k;
        }
        std::vector<uint8_t> out(262144);
        const uint64_t in_off = 0, in_len = tail.size(), out_off = 0, out_cap = out.size();
        uint64_t out_len = 0, in_used = 0;
        uint32_t det[2] = {0, 0}, chunks = 0;
        // (an empty tail: in_base may be NULL when no decoder has input)
        const uint32_t idx0 = 0;  // (decoder 0 of the pool; idx = NULL would mean all of them)
        const int rc = pzg_decoder_feed(dec, &idx0, 1, tail.empty() ? nullptr : tail.data(), &in_off, &in_len, nullptr, out.data(), &out_off,
                                        &out_cap, &out_len, &st, det, &in_used, &chunks, nullptr);
        CHECK(rc == PZG_RC_OK);
        got.insert(got.end(), out.begin(), out.begin() + out_len);
        tail.erase(tail.begin(), tail.begin() + in_used);
    }
    CHECK(st == PZG_OK);
    CHECK(got == gold);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::vector<uint8_t> z = slurp(argv[1]), gold = slurp(argv[2]);

    // 1. decoder destroyed first, then the context (the order round 2 supported)
    {
        pzg_ctx *ctx = nullptr;
        CHECK(pzg_init(0, &ctx) == PZG_RC_OK);
        pzg_decoder *d = nullptr;
        CHECK(pzg_decoder_create(ctx, 2, &d) == PZG_RC_OK);
        CHECK(run_stream(d, z, gold, 7000) == 0);
        pzg_decoder_destroy(d);
        pzg_shutdown(ctx);
    }
    // 2. pzg_shutdown first; the decoders keep working; the last destroy frees the context (round 2: use after free)
    {
        pzg_ctx *ctx = nullptr;
        CHECK(pzg_init(0, &ctx) == PZG_RC_OK);
        pzg_decoder *d1 = nullptr, *d2 = nullptr;
        CHECK(pzg_decoder_create(ctx, 1, &d1) == PZG_RC_OK);
        CHECK(pzg_decoder_create(ctx, 3, &d2) == PZG_RC_OK);
        pzg_shutdown(ctx);
        // the invalidated handle is refused for as long as a decoder keeps the context alive
        pzg_decoder *d3 = nullptr;
        CHECK(pzg_decoder_create(ctx, 1, &d3) == PZG_RC_BAD_ARG && d3 == nullptr);
        CHECK(pzg_sync(ctx) == PZG_RC_BAD_ARG);
        uint64_t out_len = 0;
        int32_t status = 0;
        uint8_t byte = 0;
        CHECK(pzg_decompress(ctx, z.data(), z.size(), &byte, 1, &out_len, &status, nullptr, nullptr) == PZG_RC_BAD_ARG);
        pzg_shutdown(ctx);  // a second shutdown: ignored
        CHECK(run_stream(d1, z, gold, 997) == 0);
        pzg_decoder_destroy(d1);
        CHECK(pzg_decoder_reset(d2, nullptr, 0) == PZG_RC_OK);
        CHECK(run_stream(d2, z, gold, 1 << 20) == 0);
        pzg_decoder_destroy(d2);  // frees the context
    }
    // 3. NULLs are ignored; a fresh context after all of that still decodes
    {
        pzg_decoder_destroy(nullptr);
        pzg_shutdown(nullptr);
        pzg_ctx *ctx = nullptr;
        CHECK(pzg_init(0, &ctx) == PZG_RC_OK);
        std::vector<uint8_t> out(gold.size() + 16);
        uint64_t out_len = 0, in_used = 0;
        int32_t status = -1;
        uint32_t det[2];
        CHECK(pzg_decompress(ctx, z.data(), z.size(), out.data(), gold.size(), &out_len, &status, det, &in_used) == PZG_RC_OK);
        CHECK(status == PZG_OK && out_len == gold.size() && memcmp(out.data(), gold.data(), gold.size()) == 0);
        pzg_shutdown(ctx);
    }
    printf("lifetimes ok\n");
    return 0;
}
