// test_mirror.cpp -- the reference's test/Test.hs (:56-86, the nine .z/.gold cases) plus the behaviours
// SURVEY.md 8a pins by reading, written against the C++ module mirror
// (pure_zlib_amd/cxx/codec_compression_zlib.hpp) the way the Haskell tests are written against
// Codec.Compression.Zlib.  Usage: test_mirror <dir with name.z/name.gold pairs>.  Needs a GPU.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../pure_zlib_amd/cxx/codec_compression_zlib.hpp"

using namespace Codec::Compression::Zlib;

static ByteString readFile(const std::string &path)
{
    ByteString s;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path.c_str());
        exit(2);
    }
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

static int failures = 0;
#define CHECK(cond, name)                                      \
    do {                                                       \
        const bool ok_ = (cond);                               \
        printf("%-62s %s\n", name, ok_ ? "OK" : "FAILED");     \
        if (!ok_) ++failures;                                  \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const char *cases[] = {"randtest1", "randtest2", "randtest3", "rfctest1", "rfctest2", "rfctest3", "zerotest1", "zerotest2", "zerotest3"};

    // Test.hs:56-86: decompress (L.readFile x.z) == Right (L.readFile x.gold)
    std::vector<LazyByteString> all;
    std::vector<ByteString> golds;
    for (const char *tc : cases) {
        const ByteString z = readFile(dir + "/" + tc + ".z"), gold = readFile(dir + "/" + tc + ".gold");
        const Either r = decompress(fromChunksOf(z));
        CHECK(r.is_right && r.right == gold, (std::string("decompress ") + tc + ".z == " + tc + ".gold").c_str());
        all.push_back(fromChunksOf(z));
        golds.push_back(gold);
    }
    // decompressMany: the same nine in one launch
    {
        const std::vector<Either> rs = decompressMany(all);
        bool ok = rs.size() == all.size();
        for (size_t i = 0; ok && i < rs.size(); ++i) ok = rs[i].is_right && rs[i].right == golds[i];
        CHECK(ok, "decompressMany [nine cases] == map Right golds");
    }
    const ByteString z1 = readFile(dir + "/rfctest1.z"), g1 = readFile(dir + "/rfctest1.gold");
    // Zlib.hs:46-49: trailing bytes inside the last chunk are ignored, a whole unread chunk is an error
    {
        const Either a = decompress(fromStrict(z1 + "garbage"));
        CHECK(a.is_right && a.right == g1, "trailing bytes inside the last chunk are ignored");
        const Either b = decompress(LazyByteString{z1, "garbage"});
        CHECK(!b.is_right && b.left.show() == "Decompression error: Finished with data remaining.",
              "a whole unread chunk: Left \"Finished with data remaining.\"");
    }
    // error values: constructor and the exact Show text (Monad.hs:95-102 and the raise sites)
    {
        const Either t = decompress(fromStrict(z1.substr(0, z1.size() / 2)));
        CHECK(!t.is_right && t.left.constructor == DecompressionError::DecompressionError_ &&
                  t.left.show() == "Decompression error: Ran out of data mid-decompression 2.",
              "truncated input: DecompressionError \"Ran out of data mid-decompression 2.\"");
        const Either h = decompress(fromStrict(ByteString("\x78\x9d\x00", 3)));
        CHECK(!h.is_right && h.left.constructor == DecompressionError::HeaderError && h.left.show() == "Header error: Header checksum failed",
              "bad FCHECK: HeaderError \"Header checksum failed\"");
        ByteString bad = z1;
        bad[bad.size() - 1] ^= 0x55;
        const Either c = decompress(fromStrict(bad));
        CHECK(!c.is_right && c.left.constructor == DecompressionError::ChecksumError &&
                  c.left.show().rfind("Checksum error: checksum mismatch: ", 0) == 0,
              "corrupt trailer: ChecksumError \"checksum mismatch: <theirs> != <ours>\"");
        DecompressionError e1 = t.left, e2 = t.left;
        e2.status = 0;  // deriving Eq looks at constructor and message only
        CHECK(e1 == e2 && e1 != h.left, "DecompressionError: deriving Eq");
    }
    // decompressIncremental (Deflate.hs:30-48 drives it like this): NeedMore* -> Chunk* -> Done
    {
        const ByteString z3 = readFile(dir + "/zerotest3.z"), g3 = readFile(dir + "/zerotest3.gold");
        ZlibDecoder d = decompressIncremental();
        bool ok = d.state() == ZlibDecoder::NeedMore;
        d.feed("");
        ok = ok && d.state() == ZlibDecoder::NeedMore;
        const LazyByteString pieces = fromChunksOf(z3, 300);
        ByteString got;
        std::vector<size_t> sizes;
        size_t early = 0;  // chunks handed out before the last piece was fed: the decoder really is incremental
        for (size_t i = 0; i < pieces.size(); ++i) {
            ok = ok && d.state() == ZlibDecoder::NeedMore;
            d.feed(pieces[i]);
            while (d.state() == ZlibDecoder::Chunk) {
                got += d.chunk();
                sizes.push_back(d.chunk().size());
                if (i + 1 < pieces.size()) ++early;
                d.next();
            }
        }
        ok = ok && (g3.size() < 200000 || early > 0);
        ok = ok && d.state() == ZlibDecoder::Done && got == g3;
        for (size_t i = 0; i + 1 < sizes.size(); ++i) ok = ok && sizes[i] == 32768;
        ok = ok && !sizes.empty() && sizes.back() >= 32768 && sizes.back() < 65536;
        CHECK(ok, "decompressIncremental: NeedMore*, 32768-byte Chunks + the rest, Done");
        ZlibDecoder e = decompressIncremental();
        e.feed(ByteString("\x78\x9d\x00", 3));
        CHECK(e.state() == ZlibDecoder::DecompError && e.error().show() == "Header error: Header checksum failed",
              "decompressIncremental: DecompError (HeaderError ...)");
    }
    printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
