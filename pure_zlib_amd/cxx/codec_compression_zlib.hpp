// codec_compression_zlib.hpp -- the reference's module surface in C++, over the C ABI of include/pzg.h.
//
// pure-zlib is compiled (Haskell) code whose boundary is a module signature, not an FFI
// (src/Codec/Compression/Zlib.hs:3-8).  No GHC exists in this project's image, so next to the Haskell
// shim of INTEGRATION.md (source only) this header is the host side that is actually compiled and run:
// same names, same argument meaning, same error behaviour, header-only, C++17, no dependency beyond
// libpzg.so.  There is no CPU decode path here either: construction fails without a GPU.
//
//   Codec.Compression.Zlib                      namespace Codec::Compression::Zlib
//   ----------------------------------------    -----------------------------------------------
//   data DecompressionError (Monad.hs:87-104)   struct DecompressionError { constructor, message; show(); == }
//   L.ByteString (lazy = list of chunks)        using LazyByteString = std::vector<std::string>
//   decompress      (Zlib.hs:32-51)             Either decompress(const LazyByteString&)
//   decompressMany  (new)            std::vector<Either> decompressMany(const std::vector<LazyByteString>&)
//   ZlibDecoder / decompressIncremental         class ZlibDecoder { NeedMore / Chunk / Done / DecompError }
//     (Monad.hs:163-167, Zlib.hs:29-30)
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pzg.h"

namespace Codec {
namespace Compression {
namespace Zlib {

using ByteString = std::string;                   // strict bytes
using LazyByteString = std::vector<ByteString>;   // lazy bytes: the chunk list (never holds empty chunks)

inline LazyByteString fromStrict(const ByteString &b) { return b.empty() ? LazyByteString{} : LazyByteString{b}; }
inline ByteString toStrict(const LazyByteString &l)
{
    ByteString s;
    for (const auto &c : l) s += c;
    return s;
}
// L.readFile hands out defaultChunkSize (32 KiB - overhead; 32 KiB here) pieces
inline LazyByteString fromChunksOf(const ByteString &b, size_t chunk = 32768)
{
    LazyByteString l;
    for (size_t i = 0; i < b.size(); i += chunk) l.push_back(b.substr(i, chunk));
    return l;
}

// ---- Monad.hs:87-104 ---------------------------------------------------------------------------
struct DecompressionError {
    enum Constructor {
        HuffmanTreeError,
        FormatError,
        DecompressionError_,  // the constructor that shares the type's name
        HeaderError,
        ChecksumError,
        ReferenceThrows  // inputs on which the reference throws a Haskell exception; a value here
    };
    Constructor constructor = DecompressionError_;
    std::string message;
    int32_t status = -1;  // the ABI's per-stream status this was rebuilt from
    uint32_t detail[2] = {0, 0};

    // instance Show (Monad.hs:95-102)
    std::string show() const
    {
        static const char *prefix[] = {"Huffman tree manipulation error: ", "Block format error: ", "Decompression error: ",
                                       "Header error: ", "Checksum error: ", ""};
        return std::string(prefix[constructor]) + message;
    }
    // deriving Eq
    bool operator==(const DecompressionError &o) const { return constructor == o.constructor && message == o.message; }
    bool operator!=(const DecompressionError &o) const { return !(*this == o); }
};

// Either DecompressionError L.ByteString
struct Either {
    bool is_right = false;
    DecompressionError left;
    ByteString right;  // (strict: the concatenated output; lazy-ByteString equality ignores chunking)
    static Either Left(DecompressionError e)
    {
        Either r;
        r.left = std::move(e);
        return r;
    }
    static Either Right(ByteString b)
    {
        Either r;
        r.is_right = true;
        r.right = std::move(b);
        return r;
    }
};

namespace detail {

inline DecompressionError::Constructor constructor_of(int32_t status)
{
    switch (status) {
    case PZG_E_TRUNCATED:
    case PZG_E_DATA_REMAINING: return DecompressionError::DecompressionError_;
    case PZG_E_HDR_FCHECK:
    case PZG_E_HDR_METHOD:
    case PZG_E_HDR_WINDOW: return DecompressionError::HeaderError;
    case PZG_E_FMT_LEN_NLEN:
    case PZG_E_FMT_BTYPE: return DecompressionError::FormatError;
    case PZG_E_HUFF_BUILD:
    case PZG_E_HUFF_EMPTY_TREE:
    case PZG_E_HUFF_EMPTY_BRANCH: return DecompressionError::HuffmanTreeError;
    case PZG_E_CHECKSUM: return DecompressionError::ChecksumError;
    default: return DecompressionError::ReferenceThrows;
    }
}

// rebuild the reference's error value (constructor + exact message) from (status, detail)
inline DecompressionError error_from_status(const ByteString &stream, int32_t status, const uint32_t det[2])
{
    DecompressionError e;
    e.constructor = constructor_of(status);
    e.status = status;
    e.detail[0] = det[0];
    e.detail[1] = det[1];
    char buf[256];
    pzg_error_message((const uint8_t *)stream.data(), stream.size(), status, det, buf, sizeof buf);
    const std::string text(buf);
    DecompressionError p = e;  // p.show() with an empty message is just the constructor's prefix
    p.message.clear();
    const std::string prefix = p.show();
    e.message = (!prefix.empty() && text.compare(0, prefix.size(), prefix) == 0) ? text.substr(prefix.size()) : text;
    return e;
}

inline uint64_t align16(uint64_t x) { return (x + 15u) & ~(uint64_t)15u; }

}  // namespace detail

// One pzg_ctx (HIP device + stream + staging arenas); the library locks it internally.
class Context {
  public:
    explicit Context(int device = 0)
    {
        int rc = pzg_init(device, &h_);
        if (rc != PZG_RC_OK) throw std::runtime_error(std::string("pzg_init: ") + pzg_strerror(rc) + " (there is no CPU fallback)");
    }
    ~Context()
    {
        if (h_) pzg_shutdown(h_);
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    pzg_ctx *handle() const { return h_; }

    static Context &shared()
    {
        static Context c(0);
        return c;
    }

  private:
    pzg_ctx *h_ = nullptr;
};

// Zlib.hs:46-49: `Done` with whole unread chunks left is Left "Finished with data remaining."; trailing
// bytes inside the last chunk handed over are silently ignored.
inline Either apply_chunk_rule(const LazyByteString &chunks, uint64_t in_used, ByteString data)
{
    uint64_t cum = 0;
    size_t loaded = 0;
    for (const auto &c : chunks) {
        if (cum >= in_used) break;
        cum += c.size();
        ++loaded;
    }
    if (loaded < chunks.size()) {
        DecompressionError e;
        e.constructor = DecompressionError::DecompressionError_;
        e.message = "Finished with data remaining.";
        e.status = PZG_E_DATA_REMAINING;
        return Either::Left(e);
    }
    return Either::Right(std::move(data));
}

namespace detail {
// The mirrors' page-locked arenas (include/pzg.h PZG_HOST_PINNED): one input and one output buffer per thread from
// pzg_host_alloc, grow-only.  A batch is packed straight into them, so the library stages nothing a second time and the
// copy engines read / write the caller's memory at link speed.  Where the system will not lock the memory the call goes
// through ordinary vectors and the staged path: same results.
struct PinnedArena {
    uint8_t *p = nullptr;
    size_t cap = 0;
    ~PinnedArena() { pzg_host_free(p); }
    uint8_t *reserve(size_t bytes)
    {
        if (cap >= bytes) return p;
        pzg_host_free(p);
        cap = bytes + bytes / 4 < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
        p = (uint8_t *)pzg_host_alloc(cap);
        if (!p) cap = 0;
        return p;
    }
};
inline PinnedArena &arena(int which)
{
    static thread_local PinnedArena a[2];
    return a[which];
}
}  // namespace detail

// decompressMany: every stream decoded by its own wavefront in one launch.  zlib streams do not carry
// their decoded size, so each stream gets a capacity (size_hint[i], or a guess) and the streams that
// report PZG_E_OUT_TOO_SMALL are relaunched once with the exact size the kernel measured.
inline std::vector<Either> decompressMany(const std::vector<LazyByteString> &inputs, Context &ctx = Context::shared(),
                                          const std::vector<uint64_t> *size_hint = nullptr)
{
    const size_t n = inputs.size();
    std::vector<Either> results(n);
    std::vector<ByteString> flat(n);
    std::vector<uint64_t> caps(n);
    for (size_t i = 0; i < n; ++i) {
        flat[i] = toStrict(inputs[i]);
        caps[i] = size_hint ? (*size_hint)[i] : (flat[i].size() * 4 > 256 ? flat[i].size() * 4 : 256);  // a modest first guess: too-small streams are relaunched with their exact size
    }
    std::vector<size_t> todo(n);
    for (size_t i = 0; i < n; ++i) todo[i] = i;
    for (int attempt = 0; attempt < 2 && !todo.empty(); ++attempt) {
        const size_t m = todo.size();
        std::vector<uint64_t> in_off(m), in_len(m), out_off(m), out_cap(m), out_len(m), in_used(m);
        std::vector<int32_t> status(m, -1);
        std::vector<uint32_t> det(2 * m), adler(m);
        uint64_t ipos = 0, opos = 0;
        for (size_t k = 0; k < m; ++k) {  // 16-byte aligned extents: the wide store path
            in_off[k] = ipos;
            out_off[k] = opos;
            in_len[k] = flat[todo[k]].size();
            out_cap[k] = caps[todo[k]];
            ipos += detail::align16(in_len[k]);
            opos += detail::align16(out_cap[k]);
        }
        // packed at ascending offsets into the thread's page-locked arenas (PZG_HOST_PINNED); pageable vectors if those are refused
        uint8_t *in_p = detail::arena(0).reserve(ipos + 16), *out_p = detail::arena(1).reserve(opos + 16);
        const bool pinned = in_p && out_p;
        std::vector<uint8_t> in_vec, out_vec;
        if (!pinned) {
            in_vec.resize(ipos + 16);
            out_vec.resize(opos + 16);
            in_p = in_vec.data();
            out_p = out_vec.data();
        }
        for (size_t k = 0; k < m; ++k)
            if (in_len[k]) memcpy(in_p + in_off[k], flat[todo[k]].data(), in_len[k]);
        int rc = pzg_decompress_many(ctx.handle(), in_p, in_off.data(), in_len.data(), out_p, out_off.data(),
                                     out_cap.data(), out_len.data(), status.data(), det.data(), in_used.data(), adler.data(),
                                     (uint32_t)m, pinned ? PZG_HOST_PINNED : 0u);
        if (rc != PZG_RC_OK) throw std::runtime_error(std::string("pzg_decompress_many: ") + pzg_last_error(ctx.handle()));
        std::vector<size_t> retry;
        for (size_t k = 0; k < m; ++k) {
            const size_t i = todo[k];
            if (status[k] == PZG_OK) {
                results[i] = apply_chunk_rule(inputs[i], in_used[k], ByteString((const char *)out_p + out_off[k], out_len[k]));
            } else if (status[k] == PZG_E_OUT_TOO_SMALL && attempt == 0) {
                caps[i] = out_len[k];
                retry.push_back(i);
            } else {
                results[i] = Either::Left(detail::error_from_status(flat[i], status[k], &det[2 * k]));
            }
        }
        todo.swap(retry);
    }
    return results;
}

// Codec.Compression.Zlib.decompress (Zlib.hs:32-51): pure, strict, same Left values
inline Either decompress(const LazyByteString &ifile, Context &ctx = Context::shared())
{
    return decompressMany({ifile}, ctx)[0];
}

// ---- ZlibDecoder (Monad.hs:163-167) and decompressIncremental (Zlib.hs:29-30) -------------------------
//   data ZlibDecoder s = NeedMore (ByteString -> ST s (ZlibDecoder s)) | Chunk ByteString (ST s (ZlibDecoder s))
//                      | Done | DecompError DecompressionError
// The suspended decoder lives on the device (pzg_decoder, include/pzg.h): feed() is one launch that continues it from
// where it stopped -- nothing is re-decoded -- and reports how many 32,768-byte chunks the reference has published by
// then (moveWindow after every match and block end once 64 KiB are buffered: Monad.hs:338-347, OutputWindow.hs:45-54),
// so Chunks, NeedMore, Done and DecompError come out in exactly the reference's order.  No CPU inflate anywhere.
class ZlibDecoder {
  public:
    enum State { NeedMore, Chunk, Done, DecompError };
    State state() const { return state_; }

    // NeedMore f: apply f to the next input chunk
    void feed(const ByteString &chunk)
    {
        if (state_ != NeedMore) throw std::logic_error("feed: the decoder is not in NeedMore");
        if (chunk.empty()) return;  // S.uncons = Nothing: ask again (Monad.hs:185-197)
        tail_ += chunk;
        all_ += chunk;
        for (;;) {
            std::vector<uint8_t> out(kRoom);
            const uint64_t in_off = 0, in_len = tail_.size(), out_off = 0, out_cap = kRoom;
            uint64_t out_len = 0, in_used = 0;
            int32_t st = 0;
            uint32_t det[2] = {0, 0}, chunks = 0;
            static const uint8_t none = 0;
            const int rc = pzg_decoder_feed(dec_.get(), nullptr, 1, tail_.empty() ? &none : (const uint8_t *)tail_.data(), &in_off, &in_len,
                                            nullptr, out.data(), &out_off, &out_cap, &out_len, &st, det, &in_used, &chunks, nullptr);
            if (rc != PZG_RC_OK) throw std::runtime_error(std::string("pzg_decoder_feed: ") + pzg_strerror(rc));
            pending_.append((const char *)out.data(), out_len);
            tail_.erase(0, in_used);
            device_chunks_ = chunks;
            if (st == PZG_DEC_OUT_FULL) continue;  // out of room: the rest of the input, fresh room
            if (st == PZG_DEC_NEED_INPUT) {
                terminal_ = NeedMore;
            } else if (st == PZG_OK) {
                terminal_ = Done;
            } else {
                terminal_ = DecompError;
                error_ = detail::error_from_status(all_, st, det);
            }
            break;
        }
        advance();
    }
    // Chunk c m: the chunk, then run the continuation m
    const ByteString &chunk() const { return cur_; }
    void next()
    {
        if (state_ != Chunk) throw std::logic_error("next: the decoder is not in Chunk");
        advance();
    }
    const DecompressionError &error() const { return error_; }

    // The decoder takes its own reference on the library's context (include/pzg.h "Lifetimes"): like the reference's
    // closure (Monad.hs:163-167) it may outlive the Context object it was made from and be dropped in any order.
    explicit ZlibDecoder(Context &ctx = Context::shared())
    {
        pzg_decoder *d = nullptr;
        const int rc = pzg_decoder_create(ctx.handle(), 1, &d);
        if (rc != PZG_RC_OK) throw std::runtime_error(std::string("pzg_decoder_create: ") + pzg_strerror(rc));
        dec_ = std::shared_ptr<pzg_decoder>(d, pzg_decoder_destroy);
    }

  private:
    static constexpr size_t kExcess = 32768;    // OutputWindow.hs:42-43 excessChunkSize
    static constexpr size_t kRoom = 256 * 1024;  // output room per launch
    // the next constructor: the chunks moveWindow has published so far, then what the last feed ended in
    void advance()
    {
        if (published_ < device_chunks_) {
            cur_ = pending_.substr(0, kExcess);
            pending_.erase(0, kExcess);
            ++published_;
            state_ = Chunk;
        } else if (terminal_ == Done && !final_published_) {  // finalize (Monad.hs:349-353): whatever is left, as one chunk
            cur_ = pending_;
            pending_.clear();
            final_published_ = true;
            state_ = Chunk;
        } else {
            state_ = terminal_;
        }
    }
    std::shared_ptr<pzg_decoder> dec_;
    State state_ = NeedMore, terminal_ = NeedMore;
    ByteString tail_, all_, pending_, cur_;
    uint32_t published_ = 0, device_chunks_ = 0;
    bool final_published_ = false;
    DecompressionError error_;
};

inline ZlibDecoder decompressIncremental(Context &ctx = Context::shared()) { return ZlibDecoder(ctx); }

}  // namespace Zlib
}  // namespace Compression
}  // namespace Codec
