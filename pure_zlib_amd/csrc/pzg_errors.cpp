// pzg_errors.cpp -- pzg_error_message(): the exact `show` text of the reference's
// DecompressionError (src/Codec/Compression/Zlib/Monad.hs:95-102) for a per-stream status.
//
// Every status but PZG_E_HUFF_BUILD carries its message in (status, detail).  For
// PZG_E_HUFF_BUILD the reference's text depends on the order createHuffmanTree inserts the codes
// (HuffmanTree.hs:25-34: foldr, i.e. LAST symbol first), so this re-reads the one dynamic-block
// header the kernel pointed at (detail[1] = its bit offset) and replays the insertions.  That is
// header parsing only (a few hundred bits); no stream is ever inflated on the CPU.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pzg.h"

namespace {

struct Bits {
    const uint8_t *p;
    uint64_t nbits, pos;
    bool ok = true;
    uint32_t get(int n)
    {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) {
            if (pos >= nbits) {
                ok = false;
                return 0;
            }
            v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1u) << i;
            pos++;
        }
        return v;
    }
};

struct Code {
    int sym, len;
    uint32_t code;
};

// Deflate.hs:261-288 computeCodeValues: canonical codes in ascending symbol order, zero lengths dropped
int canonical(const int *lens, int n, Code *out)
{
    uint32_t bl[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; ++i)
        if (lens[i]) bl[lens[i]]++;
    uint32_t code = 0;
    for (int b = 1; b < 16; ++b) {
        code = (code + bl[b - 1]) << 1;
        next[b] = code;
    }
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (lens[i]) {
            out[m].sym = i;
            out[m].len = lens[i];
            out[m].code = next[lens[i]]++;
            m++;
        }
    return m;
}

// HuffmanTree.hs:25-71: insert from the last triple to the first; report the first failing insertion.
// An insertion of (c, l) fails iff an earlier one is a proper prefix of it ("HuffmanValue hit ..."),
// equals it ("Two values ..."), or it is a proper prefix of an earlier one ("... leaf is a node").
bool replay_insertions(const Code *c, int m, char *buf, size_t cap)
{
    for (int i = m - 1; i >= 0; --i) {
        const int l = c[i].len;
        const uint32_t ci = c[i].code & ((1u << l) - 1u);
        int kind = -1;
        for (int j = m - 1; j > i && kind < 0; --j) {
            const int lj = c[j].len;
            const uint32_t cj = c[j].code & ((1u << lj) - 1u);
            if (lj < l && cj == (ci >> (l - lj))) kind = 1;
        }
        for (int j = m - 1; j > i && kind < 0; --j)
            if (c[j].len == l && (c[j].code & ((1u << l) - 1u)) == ci) kind = 0;
        for (int j = m - 1; j > i && kind < 0; --j) {
            const int lj = c[j].len;
            const uint32_t cj = c[j].code & ((1u << lj) - 1u);
            if (lj > l && (cj >> (lj - l)) == ci) kind = 2;
        }
        if (kind == 0) {
            snprintf(buf, cap, "Huffman tree manipulation error: Two values point to the same place!");
            return true;
        }
        if (kind == 1) {
            snprintf(buf, cap, "Huffman tree manipulation error: HuffmanValue hit while inserting a value!");
            return true;
        }
        if (kind == 2) {
            snprintf(buf, cap, "Huffman tree manipulation error: Tried to add where the leaf is a node: %d", c[i].sym);
            return true;
        }
    }
    return false;
}

// one symbol of a (valid, possibly incomplete) canonical code, bit by bit
int decode_sym(Bits &b, const Code *c, int m)
{
    uint32_t acc = 0;
    for (int l = 1; l < 16; ++l) {
        acc = (acc << 1) | b.get(1);
        if (!b.ok) return -1;
        for (int i = 0; i < m; ++i)
            if (c[i].len == l && c[i].code == acc) return c[i].sym;
    }
    return -1;
}

bool explain_huff_build(const uint8_t *in, uint64_t in_len, uint32_t tree, uint32_t bit_off, char *buf, size_t cap)
{
    static const int ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits b{in, in_len * 8u, bit_off};
    (void)b.get(1);
    if (b.get(2) != 2u || !b.ok) return false;
    const int hlit = 257 + (int)b.get(5), hdist = 1 + (int)b.get(5), hclen = 4 + (int)b.get(4);
    int cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[ORDER[i]] = (int)b.get(3);
    if (!b.ok) return false;
    Code cc[19];
    const int mc = canonical(cl, 19, cc);
    if (tree == PZG_TREE_CODELEN) return replay_insertions(cc, mc, buf, cap);
    int lens[288 + 32 + 138 + 8] = {0};
    int n = 0, prev = 0;
    while (n < hlit + hdist) {  // Deflate.hs:124-156
        const int s = decode_sym(b, cc, mc);
        if (s < 0) return false;
        if (s <= 15) {
            lens[n++] = s;
            prev = s;
        } else {
            int num, val;
            if (s == 16) {
                num = 3 + (int)b.get(2);
                val = prev;
            } else if (s == 17) {
                num = 3 + (int)b.get(3);
                val = prev = 0;
            } else {
                num = 11 + (int)b.get(7);
                val = prev = 0;
            }
            for (int k = 0; k < num; ++k) lens[n + k] = val;
            n += num;
        }
        if (!b.ok) return false;
    }
    Code codes[320 + 138 + 8];
    if (tree == PZG_TREE_LITLEN) return replay_insertions(codes, canonical(lens, hlit, codes), buf, cap);
    return replay_insertions(codes, canonical(lens + hlit, n - hlit, codes), buf, cap);
}

}  // namespace

extern "C" int pzg_error_message(const uint8_t *in, uint64_t in_len, int32_t status, const uint32_t detail[2],
                                 char *buf, size_t buf_len)
{
    if (!buf || buf_len == 0) return 0;
    const uint32_t d0 = detail ? detail[0] : 0, d1 = detail ? detail[1] : 0;
    buf[0] = 0;
    switch (status) {
    case PZG_OK: break;
    case PZG_E_TRUNCATED: snprintf(buf, buf_len, "Decompression error: Ran out of data mid-decompression 2."); break;
    case PZG_E_HDR_FCHECK: snprintf(buf, buf_len, "Header error: Header checksum failed"); break;
    case PZG_E_HDR_METHOD: snprintf(buf, buf_len, "Header error: Bad compression method: %u", d0); break;
    case PZG_E_HDR_WINDOW: snprintf(buf, buf_len, "Header error: Window size too big: %u", d0); break;
    case PZG_E_FMT_LEN_NLEN: snprintf(buf, buf_len, "Block format error: Len/nlen mismatch in uncompressed block."); break;
    case PZG_E_FMT_BTYPE: snprintf(buf, buf_len, "Block format error: Unacceptable BTYPE: 3"); break;
    case PZG_E_HUFF_BUILD:
        if (!in || !explain_huff_build(in, in_len, d0, d1, buf, buf_len))
            snprintf(buf, buf_len, "Huffman tree manipulation error: (over-subscribed code)");
        break;
    case PZG_E_HUFF_EMPTY_TREE: snprintf(buf, buf_len, "Huffman tree manipulation error: Tried to advance empty tree!"); break;
    case PZG_E_HUFF_EMPTY_BRANCH: snprintf(buf, buf_len, "Huffman tree manipulation error: Advanced to empty tree!"); break;
    case PZG_E_CHECKSUM: snprintf(buf, buf_len, "Checksum error: checksum mismatch: %x != %x", d0, d1); break;
    case PZG_E_BAD_DISTANCE:
        snprintf(buf, buf_len, "(reference throws) back-reference distance %u exceeds the %u bytes produced", d0, d1);
        break;
    case PZG_E_BAD_LITLEN_SYMBOL:
        snprintf(buf, buf_len, "(reference throws) literal/length symbol %u has no length entry", d0);
        break;
    case PZG_E_BAD_DIST_SYMBOL:
        snprintf(buf, buf_len, "(reference throws) distance symbol %u has no distance entry", d0);
        break;
    case PZG_E_OUT_TOO_SMALL: snprintf(buf, buf_len, "(not a reference outcome) output buffer too small"); break;
    case PZG_E_DATA_REMAINING: snprintf(buf, buf_len, "Decompression error: Finished with data remaining."); break;
    case PZG_E_GZIP_HEADER:  // extension (RFC 1952): no reference text exists for these
        if (d0 == 1) snprintf(buf, buf_len, "Header error: gzip: bad magic");
        else if (d0 == 2) snprintf(buf, buf_len, "Header error: gzip: bad compression method: %u", d1);
        else if (d0 == 3) snprintf(buf, buf_len, "Header error: gzip: reserved flag bits set");
        else snprintf(buf, buf_len, "Header error: gzip: header crc mismatch");
        break;
    case PZG_E_GZIP_ISIZE: snprintf(buf, buf_len, "Checksum error: gzip: length mismatch: %u != %u", d0, d1); break;
    case PZG_E_DICT: snprintf(buf, buf_len, "Header error: preset dictionary mismatch: %x != %x", d0, d1); break;  // extension
    default: snprintf(buf, buf_len, "unknown status %d", status); break;
    }
    return (int)strlen(buf);
}
