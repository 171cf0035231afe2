/*
 * pzg.h -- C ABI of the MI355X-native batched zlib/DEFLATE decompressor.
 *
 * This is the drop-in boundary for the one hot path of GaloisInc/pure-zlib:
 *
 *     Codec.Compression.Zlib.decompress :: L.ByteString -> Either DecompressionError L.ByteString
 *         (reference: src/Codec/Compression/Zlib.hs:32-51)
 *
 * The reference is pure Haskell with no FFI of its own; these entry points are
 * what a `foreign import ccall safe` in a replacement Codec.Compression.Zlib
 * module binds (the binding a maintainer would add is shown in INTEGRATION.md).
 * Plain pointers and sizes only; no C++/torch/HIP types cross the boundary.
 *
 * There is NO CPU backend: every entry point that computes runs hand-written
 * HIP kernels on a gfx950 device and fails loudly (negative return code) when no
 * device is usable.
 *
 * Threading: a pzg_ctx may be shared by threads.  Host-pointer calls take one of the context's
 * independent pipelines (HIP streams + staging) each and overlap; device-pointer launches are
 * enqueued on the context's stream in call order, each with its own work counter.
 * Ownership: the caller allocates and owns every buffer; the library retains no
 * caller pointer past return (except with PZG_ASYNC, until pzg_sync()).
 * No exception crosses the boundary.
 *
 * Lifetimes: the reference's ZlibDecoder is a garbage-collected closure (Monad.hs:163-167,185-197): it can be dropped
 * at any time, in any order, and cannot dangle.  The same holds here.  A pzg_ctx is reference counted: the handle
 * pzg_init returns is one reference, every live pzg_decoder holds another.  pzg_shutdown() waits for the work
 * enqueued through the handle, drops the caller's reference and invalidates the handle (every later call on it fails
 * with PZG_RC_BAD_ARG instead of touching freed memory, for as long as a decoder keeps the context alive; after that
 * the handle is dangling like any freed pointer).  Decoders created from it stay fully usable after pzg_shutdown --
 * pzg_decoder_feed / _reset / _destroy in any order -- and the device resources go with the last
 * pzg_decoder_destroy().  pzg_shutdown() then pzg_decoder_destroy(), or the reverse, are both correct; finalisers
 * (GHC ForeignPtr, Python __del__, C++ shared_ptr) may therefore run in whatever order the collector picks.
 */
#ifndef PZG_H
#define PZG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: these declarations are its whole dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#define PZG_API __attribute__((visibility("default")))
#else
#define PZG_API
#endif

#define PZG_VERSION_MAJOR 0
#define PZG_VERSION_MINOR 5

/* ---- call-level return codes (the int every function returns) ---------------- */
#define PZG_RC_OK            0
#define PZG_RC_BAD_ARG      (-1)
#define PZG_RC_NO_DEVICE    (-2)  /* no gfx950 device / HIP runtime unusable: there is no CPU fallback */
#define PZG_RC_HIP_ERROR    (-3)  /* a HIP call failed; pzg_last_error() has the text */
#define PZG_RC_NO_MEMORY    (-4)

/* ---- per-stream status codes --------------------------------------------------
 * One per reference outcome.  `show` strings are those of
 * src/Codec/Compression/Zlib/Monad.hs:95-102 plus the raise site quoted. */
#define PZG_OK                    0  /* Right bytes                                              Zlib.hs:46-47  */
#define PZG_E_TRUNCATED           1  /* DecompressionError "Ran out of data mid-decompression 2." Zlib.hs:38-39  */
#define PZG_E_HDR_FCHECK          2  /* HeaderError "Header checksum failed"                      Zlib.hs:62-63  */
#define PZG_E_HDR_METHOD          3  /* HeaderError "Bad compression method: <d0>"                Zlib.hs:64-65  */
#define PZG_E_HDR_WINDOW          4  /* HeaderError "Window size too big: <d0>"                   Zlib.hs:66-67  */
#define PZG_E_FMT_LEN_NLEN        5  /* FormatError "Len/nlen mismatch in uncompressed block."    Deflate.hs:75-76 */
#define PZG_E_FMT_BTYPE           6  /* FormatError "Unacceptable BTYPE: 3"                       Deflate.hs:102-104 */
#define PZG_E_HUFF_BUILD          7  /* HuffmanTreeError <insert message>; d0 = tree id, d1 = bit offset of the block header
                                        modulo 2^32 (pzg_error_message re-derives the exact text from it: exact for streams
                                        with less than 512 MiB of compressed data in front of the block)  HuffmanTree.hs:55-63 */
#define PZG_E_HUFF_EMPTY_TREE     8  /* HuffmanTreeError "Tried to advance empty tree!"           HuffmanTree.hs:76 */
#define PZG_E_HUFF_EMPTY_BRANCH   9  /* HuffmanTreeError "Advanced to empty tree!"                HuffmanTree.hs:80 */
#define PZG_E_CHECKSUM           10  /* ChecksumError "checksum mismatch: <hex d0> != <hex d1>"   Deflate.hs:56-63 */
#define PZG_E_BAD_DISTANCE       11  /* reference THROWS (OutputWindow.hs:87 slice bounds); d0 = distance, d1 = bytes produced */
#define PZG_E_BAD_LITLEN_SYMBOL  12  /* reference THROWS (Deflate.hs:160-166 array index); d0 = symbol 286/287 */
#define PZG_E_BAD_DIST_SYMBOL    13  /* reference THROWS (Deflate.hs:199-205 array index); d0 = symbol >= 30 */
#define PZG_E_OUT_TOO_SMALL      14  /* not a reference outcome: out_len[i] holds the size needed */
#define PZG_E_DATA_REMAINING     15  /* DecompressionError "Finished with data remaining." -- produced by the host
                                        mirror from in_used[] and the caller's chunking (Zlib.hs:48-49), never by the kernel */
/* PZG_GZIP only (extension, no reference counterpart): */
#define PZG_E_GZIP_HEADER        18  /* "Header error: gzip: ..."; d0 = 1 bad magic, 2 method != 8, 3 reserved flag bits, 4 header CRC16 */
#define PZG_E_GZIP_ISIZE         19  /* "Checksum error: gzip: length mismatch: <d0> != <d1>" (ISIZE vs bytes produced mod 2^32);
                                      * a CRC-32 mismatch is PZG_E_CHECKSUM.  Not checked for PZG_E_OUT_TOO_SMALL streams. */

/* pzg_decompress_many_dict only (extension): */
#define PZG_E_DICT               20  /* "Header error: preset dictionary mismatch: <hex d0> != <hex d1>": the stream's DICTID (d0) is
                                      * not the Adler-32 of the dictionary supplied for it (d1) */
/* Reference raise sites that have NO status because they cannot fire (here or in the reference):
 *   Deflate.hs:150-151  DecompressionError "Unexpected code: <n>" -- getCodeLengths' fall-through for a code-length symbol outside
 *                       0..18; the code-length alphabet has exactly 19 symbols (Deflate.hs:87-88 builds its tree from 19 lengths),
 *                       so nextCode can only return 0..18.
 *   HuffmanTree.hs:77   HuffmanTreeError "Tried to advance value!" -- advanceTree on a leaf; nextCode (Monad.hs:295-302) restarts
 *                       from the root as soon as advanceTree returns a value, so a leaf is never advanced.
 *   Monad.hs:277        FormatError "Can't get a block on a non-byte boundary." -- nextBlock with bits pending; its only caller
 *                       (Deflate.hs:70-72) runs advanceToByte first.
 * The oracle keeps the first two as dead code for fidelity; no input reaches them. */
/* pzg_decoder_feed only: a decoder that is not finished */
#define PZG_DEC_NEED_INPUT      101  /* NeedMore (Monad.hs:164): every complete element of the input has been decoded */
#define PZG_DEC_OUT_FULL        102  /* this call's output room is used up: call again with the rest of the input */

/* tree ids in detail[2*i] of PZG_E_HUFF_BUILD */
#define PZG_TREE_CODELEN 0
#define PZG_TREE_LITLEN  1
#define PZG_TREE_DIST    2

/* ---- flags ------------------------------------------------------------------- */
#define PZG_DEVICE_PTRS  1u  /* every pointer argument is device memory on the context's device */
#define PZG_ASYNC        2u  /* enqueue only (requires PZG_DEVICE_PTRS); caller calls pzg_sync().  Launches enqueued on different
                              * streams (pzg_set_stream in between) may run at the same time, two at most: a third waits, on its own
                              * stream, for the first's token scratch (see pzg_init) */
#define PZG_GZIP         4u  /* EXTENSION (the reference has no gzip: README.md:42-50 TODO): every stream is
                              * one RFC 1952 member (gzip header, deflate, CRC-32 + ISIZE); adler[] then holds the CRC-32 */
#define PZG_LPT_ORDER    8u  /* device-pointer batches of mixed sizes: launch the longest streams (largest out_cap[]) first; the
                              * permutation is built on the device.  Host-pointer batches are always launched that way. */
#define PZG_HOST_PINNED 16u  /* host-pointer batches whose in_base / out_base arenas are PAGE-LOCKED (pzg_host_alloc, or hipHostMalloc /
                              * hipHostRegister by the caller): the copy engines read and write them directly -- no packing into
                              * staging, no copy-out; this is the path the module mirrors (`decompress` / `decompressMany`) take.
                              * Requirements: extents in ascending order on both sides (in_off[i] + in_len[i] <= in_off[i+1], the same for
                              * out_off / out_cap; PZG_RC_BAD_ARG otherwise); the whole span of the output arena from the first extent
                              * to the end of the last is written (bytes in gaps BETWEEN extents, and the bytes of an extent past its
                              * stream's out_len[i] -- a stream that failed or came out short -- are unspecified afterwards: they may
                              * hold what an earlier batch of this context decoded; bytes outside the span are untouched.  The staged
                              * path, without this flag, writes out_len[i] bytes per extent and nothing else); gaps of the input arena travel over the link, so pack tightly
                              * (16-byte aligned extents keep the wide store path).  Pageable memory under this flag is still
                              * correct, only slower.  Not combined with preset dictionaries. */

typedef struct pzg_ctx pzg_ctx;

/* Create a context on HIP device `device` (0-based).  One HIP stream + grow-only device
 * arenas for the host-pointer path.  Returns PZG_RC_NO_DEVICE when HIP has no usable device.
 * Device memory the library takes for itself, grow-only until pzg_shutdown: the kernels' scratch -- 64.8 KiB per stream-wave
 * of a launch (one wave per stream, at most the residency of the chip: 6,656 waves = 421 MiB on an MI355X), two such arenas per
 * device for PZG_DEVICE_PTRS launches and one per host-path pipeline in use (four at the most: 2.5 GiB per device in the worst
 * case; PZG_OPT_SCRATCH_BYTES bounds it) -- beside the host path's staging arenas.  A launch whose scratch cannot be allocated,
 * or only in part, still decodes: the stream-waves without a slice take the slower path that needs none. */
PZG_API int  pzg_init(int device, pzg_ctx **out);
/* decompressMany over SEVERAL devices of one node (API of Zlib.hs:32-35, batched): bit d of `device_mask`
 * selects HIP device d, 0 selects every visible device.  One call of pzg_decompress_many() with HOST pointers then
 * partitions the streams over the devices (longest-processing-time-first by capacity), one host thread, HIP stream set
 * and staging arenas per device, and every result lands in the caller's own out_off[] / status[] slots: the streams
 * are independent, so nothing crosses between devices (no collective).  PZG_DEVICE_PTRS calls of pzg_decompress_many need a
 * one-device context; data that already lives on several devices goes through pzg_decompress_many_sharded(). */
PZG_API int  pzg_init_mask(uint32_t device_mask, pzg_ctx **out);
/* The same with the devices named one by one: one shard per entry, in that order (shard k of
 * pzg_decompress_many_sharded = devices[k]).  A device may be named more than once -- several shards then share it, each
 * with its own streams, arenas and host thread (a way to keep more batches in flight on one device, and how the
 * multi-shard code is exercised on a one-GPU box). */
PZG_API int  pzg_init_devices(const int32_t *devices, uint32_t ndevices, pzg_ctx **out);
PZG_API int  pzg_device_count(pzg_ctx *ctx);  /* devices (shards) of the context */
/* Drop the caller's reference (see "Lifetimes" above): synchronises, invalidates the handle, frees everything unless
 * decoders are still alive -- then the last pzg_decoder_destroy() frees it.  NULL is ignored. */
PZG_API void pzg_shutdown(pzg_ctx *ctx);

/* Make the context launch on an existing HIP stream (e.g. a framework's current stream) instead of
 * its own.  `hip_stream` is a hipStream_t passed as void*; NULL means HIP's default (null) stream.
 * pzg_reset_stream() goes back to the context's own non-blocking stream. */
PZG_API int  pzg_set_stream(pzg_ctx *ctx, void *hip_stream);
PZG_API int  pzg_reset_stream(pzg_ctx *ctx);
PZG_API int  pzg_sync(pzg_ctx *ctx);

/* Tuning options.
 * PZG_OPT_RING_BITS: log2 of the LDS ring each stream-wave keeps of its most recent output.
 *   15  the whole 32 KiB DEFLATE window lives in LDS (OutputWindow.hs as a pure LDS ring): 4 stream-waves per CU
 *   11-14  a smaller near ring; back-references older than it are read from the stream's own, already
 *       flushed output in HBM/L2.  More stream-waves per CU; the kernel is latency-bound and scales with them.
 *   Results are bit-identical for every value.  Default: PZG_DEFAULT_RING_BITS. */
#define PZG_OPT_RING_BITS 1
#define PZG_DEFAULT_RING_BITS 11
/* PZG_OPT_HOST_THREADS: helper threads (1..256) that pack / copy out the STAGED host-pointer path and the decoders' feeds
 *   (default: the machine's hardware threads, at most 24).  Set it while no host-pointer call is running. */
#define PZG_OPT_HOST_THREADS 2
/* PZG_OPT_SCRATCH_BYTES: upper bound, per SHARD of the context, on the scratch memory the LIBRARY allocates for its inflate kernels
 *   (0, the default: no bound).  A shard is one entry of the device list the context was made with: pzg_init / pzg_init_mask name
 *   every device once, so that there the bound is per device; a pzg_init_devices list that names one physical device k times
 *   makes k shards of it, each with its own arenas -- the ceiling on that device is k x the bound -- and every pzg_decoder object
 *   adds one arena of one share (bound / 6) on top: with d decoders alive the ceiling is (k + d / 6) x the bound.  A launch's stream-waves decode long runs of input through a scratch of 64.8 KiB each -- 421 MiB for a
 *   launch that fills an MI355X (6,656 stream-waves) -- and a context keeps up to six such arenas per device, grow-only: two for
 *   device-pointer launches (overlapping launches must not share one) and one per host-path pipeline (four), 2.5 GiB at the
 *   most; a pzg_decoder object keeps one more for its feeds (at most 4,096 stream-waves: 259 MiB).  With a bound every arena gets an even share (bound / 6): only as many stream-waves as fit own a slice, the others
 *   decode by the slower path that needs no scratch; below 64.8 KiB per arena all of them do.  Results never depend on it.
 *   Takes effect launch by launch (an arena larger than its share is released when it is next used). */
#define PZG_OPT_SCRATCH_BYTES 3
/* PZG_OPT_BUNDLES (0.5): 1 (default) -- a PZG_DEVICE_PTRS launch of 32,768 or more zlib streams first takes its streams of the
 *   FIXED code (what `compressobj(1, ..., Z_FIXED)` and level-1 encoders make of short records) 64 to a wavefront, one lane per
 *   stream; every other stream, and every stream that is not plain (another block type, any error, a stream or capacity of a MiB or
 *   more), is decoded by the ordinary one-stream-per-wavefront kernel as before; a launch in which no stream turned out to be one for
 *   the bundles lets the next 2, 6, 14 ... 64 launches of the context go without looking (the look costs such a launch ~1 %).
 *   2 -- launches of any size, always looking.  0 -- off.  Results never depend on it; it needs no scratch memory.
 *   Setting the option (to the value it already has, too) ends the launches a context is going without looking. */
#define PZG_OPT_BUNDLES 4
/* PZG_OPT_PROFILE (0.5): 1 (default) -- a stream-wave of a PZG_DEVICE_PTRS launch remembers where the tokens of the last stream it
 *   decoded lay and cuts the next stream's input into pieces of equal WORK there (checked against the stream itself before it is
 *   trusted); 0 -- pieces of equal length, always.  Results never depend on it: it is there to be measured (bench.py's
 *   hetero_variant reports both).  Takes effect launch by launch. */
#define PZG_OPT_PROFILE 5
PZG_API int  pzg_set_option(pzg_ctx *ctx, int option, int64_t value);
/* The only environment variable the library reads is PZG_RING_BITS (11..15): the default of PZG_OPT_RING_BITS for
 * contexts created afterwards. */

/* Page-locked host memory for PZG_HOST_PINNED arenas (hipHostMalloc, portable across the node's devices).  NULL when the
 * system will not lock that much.  Free with pzg_host_free (NULL is ignored).  No context is needed. */
PZG_API void *pzg_host_alloc(size_t bytes);
PZG_API void  pzg_host_free(void *p);

/*
 * decompressMany: decode n independent zlib (RFC 1950) streams, one wavefront per stream.
 * Replaces n calls of Codec.Compression.Zlib.decompress (Zlib.hs:32-51) on single-chunk inputs.
 *
 *   in_base, in_off[n], in_len[n]    stream i is in_base[in_off[i] .. in_off[i]+in_len[i])
 *   out_base, out_off[n], out_cap[n] stream i may write out_base[out_off[i] .. out_off[i]+out_cap[i])
 *   out_len[n]             bytes the stream decodes to (may exceed the capacity: PZG_E_OUT_TOO_SMALL)
 *   status[n]              PZG_OK or PZG_E_*
 *   detail[2n]             two detail words per stream (see the status table); may be NULL
 *   in_used[n]             input bytes consumed incl. the Adler trailer; may be NULL
 *   adler[n]               Adler-32 computed over the decoded bytes; may be NULL.  (Of a stream that failed: over what had been
 *                          decoded by then -- 0 if it had outgrown out_cap[i] by then: what lies past the capacity is not stored.)
 *
 * Extents may be laid out with gaps (aligned arenas) and in any order; they must not overlap on the
 * output side.  One compressed stream may be up to 16 GiB (longer ones report PZG_E_TRUNCATED); the decoded size is
 * not limited.  A 16-byte aligned out_base+out_off[i] takes the wide (16 B/lane) store path.
 *
 * Without PZG_DEVICE_PTRS all pointers are host memory and the call stages through the
 * context's device arenas (H2D, kernel, D2H) and returns when results are in host memory.
 * Return value: PZG_RC_* for the call as a whole; per-stream results are in status[].
 */
PZG_API int pzg_decompress_many(pzg_ctx *ctx,
                        const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                        uint8_t *out_base, const uint64_t *out_off, const uint64_t *out_cap,
                        uint64_t *out_len, int32_t *status, uint32_t *detail,
                        uint64_t *in_used, uint32_t *adler,
                        uint32_t n, uint32_t flags);

/*
 * decompressMany over several devices with the data ALREADY on them: a context from
 * pzg_init_mask() has one shard per device; batch b names its shard (0 .. pzg_device_count()-1) and carries pointers that
 * are all device memory of THAT shard's device, with the meaning they have in pzg_decompress_many().  Every batch is
 * enqueued on its own device's stream -- nothing crosses PCIe or xGMI, no collective: the streams are independent -- and
 * the call returns when all of them have finished (with PZG_ASYNC: at once; pzg_sync() waits for every device).
 * Several batches may name the same shard (they run one after the other on its stream).  flags: PZG_ASYNC, PZG_GZIP,
 * PZG_LPT_ORDER (PZG_DEVICE_PTRS is implied).
 */
typedef struct pzg_device_batch {
    uint32_t shard;  /* which device of the context the pointers below live on */
    uint32_t n;      /* streams in this batch */
    const uint8_t *in_base; const uint64_t *in_off, *in_len;
    uint8_t *out_base;      const uint64_t *out_off, *out_cap;
    uint64_t *out_len; int32_t *status;
    uint32_t *detail;   /* 2n or NULL */
    uint64_t *in_used;  /* n or NULL */
    uint32_t *adler;    /* n or NULL */
} pzg_device_batch;
PZG_API int pzg_decompress_many_sharded(pzg_ctx *ctx, const pzg_device_batch *batches, uint32_t nbatches, uint32_t flags);

/*
 * EXTENSION -- preset dictionaries (RFC 1950 FDICT).  The reference skips DICTID and decodes with an empty history
 * (Zlib.hs:68, a FIXME there); pzg_decompress_many does exactly that.  This entry point is the same call plus one
 * dictionary extent per stream (dict_len[i] = 0: none): a stream whose header has FDICT set and that has a
 * dictionary checks DICTID against the dictionary's Adler-32 (PZG_E_DICT) and decodes with the dictionary as the
 * history in front of its output; every other stream behaves as in pzg_decompress_many.  dict_* may be NULL.
 */
PZG_API int pzg_decompress_many_dict(pzg_ctx *ctx,
                             const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len,
                             const uint8_t *dict_base, const uint64_t *dict_off, const uint64_t *dict_len,
                             uint8_t *out_base, const uint64_t *out_off, const uint64_t *out_cap,
                             uint64_t *out_len, int32_t *status, uint32_t *detail,
                             uint64_t *in_used, uint32_t *adler,
                             uint32_t n, uint32_t flags);

/*
 * decompressIncremental / ZlibDecoder (Zlib.hs:3-8, Monad.hs:163-197; driver Deflate.hs:30-48), batched: a pzg_decoder
 * is n suspended zlib decoders living on the device (a one-device context).  pzg_decoder_feed continues the decoders
 * idx[0..m) (idx = NULL: all n, m ignored) -- one launch, one wavefront per decoder -- as far as their input and
 * output room go.  For decoder k = idx[j]:
 *   in_base + in_off[j] .. + in_len[j]   the bytes the last call did not consume (from its in_used on) followed by
 *                                        the new input; final_in[j] != 0 (array may be NULL): no more input will follow;
 *                                        in_base may be NULL when every in_len[j] is 0
 *   out_base + out_off[j] .. + out_cap[j]  room for the bytes this call delivers (out_cap >= 4096); out_len[j] of them
 *   state[j]     PZG_DEC_NEED_INPUT  suspended on input: the reference's NeedMore
 *                PZG_DEC_OUT_FULL    suspended on room: call again with in_base + in_used[j] onward
 *                PZG_OK              the stream has ended: Done (trailing input is left unconsumed)
 *                PZG_E_*             DecompError; detail[2j..] as for pzg_decompress_many
 *   in_used[j]   input bytes the decoder is done with (it remembers a partly consumed byte itself)
 *   chunks[j]    how many 32,768-byte chunks the reference has published up to this point (cumulative): moveWindow
 *                runs after every match and block end and publishes one when 64 KiB are buffered
 *                (OutputWindow.hs:45-54); the host mirror cuts the delivered bytes into exactly those Chunks, and at
 *                PZG_OK publishes the remainder as the last one (finalize, Monad.hs:349-353).
 * Host pointers only.  Nothing is re-decoded: a feed costs what its new input costs.
 * Footprint: a pzg_decoder keeps, grow-only until pzg_decoder_destroy, staging of the largest feed it has seen --
 * m x (in_len + 32) + m x out_cap bytes of page-locked host memory and as much device memory -- plus ~41 KiB of device
 * state per decoder (its registers and tables, an 8 KiB image of its wave's LDS, and the last 32 KiB it produced).  A feed
 * of 512 decoders or more that moves 64 MiB or more is pipelined in ranges (upload, launch, download and the host-side
 * copies of neighbouring ranges overlap).  Where the system will not lock that much the staging falls back to pageable memory (slower copies,
 * same results).
 */
typedef struct pzg_decoder pzg_decoder;
PZG_API int  pzg_decoder_create(pzg_ctx *ctx, uint32_t n, pzg_decoder **out);  /* takes a reference on ctx */
PZG_API void pzg_decoder_destroy(pzg_decoder *dec);  /* legal before or after pzg_shutdown(ctx); NULL is ignored */
PZG_API int  pzg_decoder_reset(pzg_decoder *dec, const uint32_t *idx, uint32_t m);  /* those decoders start a new stream */
/* (0.5) What the last LARGE pzg_decoder_feed call of `dec` (512 decoders or 64 MiB of rooms and more: the pipelined path) spent where, in
 * milliseconds: [0] the whole call; [1] packing the inputs into page-locked staging (the issuing thread); [2] waiting for the ranges'
 * kernels, [3] bringing down what their decoders delivered (a second thread); [4] copying it out into the caller's rooms (a third).
 * The three threads run side by side: the parts do not add up to the call.  -1: no such call yet. */
PZG_API int  pzg_decoder_last_feed_ms(pzg_decoder *dec, double out[5]);
PZG_API int  pzg_decoder_feed(pzg_decoder *dec, const uint32_t *idx, uint32_t m,
                      const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len, const uint8_t *final_in,
                      uint8_t *out_base, const uint64_t *out_off, const uint64_t *out_cap,
                      uint64_t *out_len, int32_t *state, uint32_t *detail, uint64_t *in_used,
                      uint32_t *chunks, uint32_t *adler);

/* decompress: the single-stream form (n = 1, host pointers). */
PZG_API int pzg_decompress(pzg_ctx *ctx, const uint8_t *in, uint64_t in_len,
                   uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                   int32_t *status, uint32_t detail[2], uint64_t *in_used);

/* Adler-32 of one buffer (Codec.Compression.Zlib.Adler32, Adler32.hs:17-57) as a device-wide
 * reduction.  `init` is a finalized Adler value (1 for a fresh checksum). */
PZG_API int pzg_adler32(pzg_ctx *ctx, const uint8_t *buf, uint64_t len, uint32_t init,
                uint32_t *out, uint32_t flags);

/* The batched form (BASELINE config 2, 262,144 x 64 KiB): out[i] = Adler-32 of base[off[i] .. off[i]+len[i]), one wave per
 * buffer.  Device memory only (PZG_DEVICE_PTRS). */
PZG_API int pzg_adler32_many(pzg_ctx *ctx, const uint8_t *base, const uint64_t *off, const uint64_t *len,
                     uint32_t *out, uint32_t n, uint32_t flags);

/* Exact `show` text of the DecompressionError the reference returns for (status, detail) on
 * this stream (host memory; needed only for PZG_E_HUFF_BUILD, whose message depends on the
 * trie insertion order).  Writes a NUL-terminated string; returns its length. */
PZG_API int pzg_error_message(const uint8_t *in, uint64_t in_len, int32_t status,
                      const uint32_t detail[2], char *buf, size_t buf_len);

/* Kernel time of the last launch on this context in milliseconds (HIP events on the launch
 * stream), or a negative value if none was recorded. */
PZG_API double pzg_last_kernel_ms(pzg_ctx *ctx);

PZG_API const char *pzg_strerror(int rc);
PZG_API const char *pzg_last_error(pzg_ctx *ctx);
PZG_API uint32_t    pzg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PZG_H */
