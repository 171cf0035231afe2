/*
 * pz_oracle.h -- CPU restatement of GaloisInc/pure-zlib's decompression path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import,
 * link or execute it, and there only as the checker (or as the timed CPU
 * "port" baseline), never as the thing shipped.  The product (libpzg.so) does
 * not link, include or call anything in this directory.
 *
 * Parity status: PINNED.  The restatement is checked against every vector the
 * reference's own test-suite holds for this path: the nine test/test-cases
 * .z/.gold pairs (test/Test.hs:56-86) and the two computeCodeValues known-answer
 * tests (test/Test.hs:13-52,107-120).  See tests/test_oracle_golden.py.
 * The reference itself (Haskell) cannot be built here (no GHC in the image), so
 * there is no oracle/_ref; error paths are pinned only by reading the source
 * (the reference has no negative tests) and are cross-checked for error CLASS
 * against system zlib where the two agree by specification.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef PZ_ORACLE_H
#define PZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Outcome codes.  Numeric values are deliberately the same as include/pzg.h's
 * PZG_* status codes so tests can compare them directly; the two headers are
 * independent files (the product never includes this one). */
enum {
    PZO_OK = 0,
    PZO_E_TRUNCATED = 1,        /* DecompressionError "Ran out of data mid-decompression 2."  Zlib.hs:38-39 */
    PZO_E_HDR_FCHECK = 2,       /* HeaderError "Header checksum failed"                        Zlib.hs:62-63 */
    PZO_E_HDR_METHOD = 3,       /* HeaderError "Bad compression method: <cm>"                  Zlib.hs:64-65 */
    PZO_E_HDR_WINDOW = 4,       /* HeaderError "Window size too big: <cinfo>"                  Zlib.hs:66-67 */
    PZO_E_FMT_LEN_NLEN = 5,     /* FormatError "Len/nlen mismatch in uncompressed block."      Deflate.hs:75-76 */
    PZO_E_FMT_BTYPE = 6,        /* FormatError "Unacceptable BTYPE: 3"                         Deflate.hs:102-104 */
    PZO_E_HUFF_BUILD = 7,       /* HuffmanTreeError <insert error>                             HuffmanTree.hs:55-63 */
    PZO_E_HUFF_EMPTY_TREE = 8,  /* HuffmanTreeError "Tried to advance empty tree!"             HuffmanTree.hs:76 */
    PZO_E_HUFF_EMPTY_BRANCH = 9,/* HuffmanTreeError "Advanced to empty tree!"                  HuffmanTree.hs:80 */
    PZO_E_CHECKSUM = 10,        /* ChecksumError "checksum mismatch: <hex> != <hex>"           Deflate.hs:56-63 */
    PZO_E_BAD_DISTANCE = 11,    /* reference THROWS (vector slice bounds, OutputWindow.hs:87)  */
    PZO_E_BAD_LITLEN_SYMBOL = 12,/* reference THROWS (Data.Array.!, Deflate.hs:161)            */
    PZO_E_BAD_DIST_SYMBOL = 13, /* reference THROWS (Data.Array.!, Deflate.hs:200)             */
    PZO_E_OUT_TOO_SMALL = 14,   /* not a reference outcome: caller's buffer too small          */
    PZO_E_DATA_REMAINING = 15,  /* DecompressionError "Finished with data remaining."          Zlib.hs:48-49 */
    PZO_E_HUFF_ADVANCE_VALUE = 16,/* HuffmanTreeError "Tried to advance value!" (unreachable)  HuffmanTree.hs:77 */
    PZO_E_BAD_CODELEN_SYMBOL = 17,/* DecompressionError "Unexpected code: n" (unreachable)      Deflate.hs:148-149 */
    /* RFC 1952 members (PZO_F_GZIP): an EXTENSION the reference does not have (README.md:42-50 lists gzip as
     * a TODO).  Restated from the RFC, pinned against system zlib (wbits = 31), not
     * against pure-zlib. */
    PZO_E_GZIP_HEADER = 18,     /* detail0: 1 magic, 2 method != 8, 3 reserved FLG bits, 4 header CRC16 */
    PZO_E_DICT = 20,            /* EXTENSION (pzo_decompress_dict): DICTID (detail0) is not the Adler-32 of the dictionary supplied (detail1) */
    PZO_E_GZIP_ISIZE = 19       /* detail0 = ISIZE in the trailer, detail1 = bytes produced mod 2^32 (CRC mismatch is PZO_E_CHECKSUM) */
};

/* detail0 of PZO_E_HUFF_BUILD: which tree failed (low byte) and which message (next byte) */
enum { PZO_TREE_CODELEN = 0, PZO_TREE_LITLEN = 1, PZO_TREE_DIST = 2 };
enum { PZO_INS_TWO_VALUES = 0, PZO_INS_VALUE_HIT = 1, PZO_INS_LEAF_IS_NODE = 2 };

/* quirk flags: things the reference would have done that the restatement reports
 * instead of doing (it stays RFC-correct). */
enum {
    PZO_QUIRK_REF_WINDOW_OVERFLOW = 1u, /* reference would throw: 128 KiB window overrun (OutputWindow.hs:64-68,74-80) */
    PZO_QUIRK_CODELEN_OVERRUN     = 2u, /* code-length repeat ran past HLIT+HDIST and was accepted (Deflate.hs:132,96-97) */
    PZO_QUIRK_REPEAT_NO_PREV      = 4u, /* code 16 with no previous length repeated 0 (Deflate.hs:91,139-142) */
    PZO_QUIRK_FDICT_SKIPPED       = 8u, /* FDICT set: DICTID skipped, empty history (Zlib.hs:68) */
    PZO_QUIRK_STOLEN_BYTE         = 16u /* nextBlock chunk-edge bug was taken (Monad.hs:280-293); only with PZO_F_REF_CHUNK_BUG */
};

/* flags for pzo_decompress_chunks */
enum {
    PZO_F_REF_CHUNK_BUG = 1u, /* replicate Monad.hs:280-293 literally (stored block ending exactly at a chunk end) */
    PZO_F_GZIP = 2u           /* the input is one RFC 1952 member: gzip header, deflate, CRC-32 + ISIZE; res->adler holds the CRC-32 */
};

typedef struct pzo_result {
    int32_t  status;      /* PZO_* */
    uint32_t detail0;
    uint32_t detail1;
    uint32_t adler;       /* Adler-32 of the bytes produced (finalizeAdler, Adler32.hs:52-57) */
    uint64_t out_len;     /* bytes produced (may exceed out_cap: count-only past the cap) */
    uint64_t in_used;     /* input bytes consumed when the decoder stopped */
    uint32_t quirks;      /* PZO_QUIRK_* */
    uint32_t n_blocks;    /* deflate blocks seen */
    char     message[192];/* `show` of the DecompressionError the reference would return, "" on success */
} pzo_result;

/* Codec.Compression.Zlib.decompress on a single strict chunk (Zlib.hs:32-51). */
int pzo_decompress(const uint8_t *in, uint64_t in_len,
                   uint8_t *out, uint64_t out_cap, pzo_result *res);

/* The same on a lazy ByteString given as n_chunks strict chunks laid end to end in `in`
 * (chunk i is in[chunk_off[i] .. chunk_off[i+1])); reproduces the per-chunk terminal-state
 * mapping of Zlib.hs:37-51 including "Finished with data remaining.". */
int pzo_decompress_chunks(const uint8_t *in, const uint64_t *chunk_off, uint32_t n_chunks,
                          uint8_t *out, uint64_t out_cap, uint32_t flags, pzo_result *res);

/* The ZlibDecoder constructors (Monad.hs:163-167) as trace events */
enum { PZO_EV_NEED_MORE = 1, PZO_EV_CHUNK = 2, PZO_EV_DONE = 3, PZO_EV_ERROR = 4 };

/* decompressIncremental driven one input piece per NeedMore (Deflate.hs:30-48): the events it goes through.
 * ev_type/ev_val[0..ev_cap) receive the first events, *n_events their total number; ev_val = chunk length / status.
 * When the pieces run out the trace ends with the NeedMore nothing answers (status PZO_E_TRUNCATED). */
int pzo_trace(const uint8_t *in, const uint64_t *chunk_off, uint32_t n_chunks, uint8_t *out, uint64_t out_cap,
              int32_t *ev_type, uint32_t *ev_val, uint32_t ev_cap, uint32_t *n_events, pzo_result *res);

/* EXTENSION: decompress with a preset dictionary (RFC 1950 FDICT).  The reference skips DICTID and decodes with an
 * empty history (Zlib.hs:68) -- that is what pzo_decompress does; this entry point installs the dictionary instead,
 * pinned against system zlib (zlib.decompressobj(zdict=...)). */
int pzo_decompress_dict(const uint8_t *in, uint64_t in_len, const uint8_t *dict, uint64_t dict_len,
                        uint8_t *out, uint64_t out_cap, pzo_result *res);

/* Adler32.hs:19-57.  `adler` is a finalized value ((b<<16)|a); pass 1 to start. */
uint32_t pzo_adler32(uint32_t adler, const uint8_t *buf, uint64_t len);

/* CRC-32 (RFC 1952 section 8; reflected 0xedb88320), `crc` = the value so far, 0 to start.  Extension, see PZO_F_GZIP. */
uint32_t pzo_crc32(uint32_t crc, const uint8_t *buf, uint64_t len);

/* Deflate.hs:261-288 computeCodeValues.  Input n (symbol,length) pairs in any order;
 * output triples ascending by symbol with zero lengths dropped.  Returns count. */
int pzo_compute_code_values(const int *syms, const int *lens, int n,
                            int *out_sym, int *out_len, int *out_code);

/* Timed-baseline helper for bench.py: decode n streams laid out like the product ABI
 * (in_off[n+1], out_off[n+1] capacities) sequentially on the calling thread.
 * Returns the number of streams whose status != PZO_OK. */
uint32_t pzo_decompress_many(const uint8_t *in_base, const uint64_t *in_off,
                             uint8_t *out_base, const uint64_t *out_off,
                             uint64_t *out_len, int32_t *status, uint32_t *adler, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
