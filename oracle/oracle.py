"""ctypes binding for the CPU oracle (oracle/pz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The product package (pure_zlib_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# mirrors pz_oracle.h
OK = 0
E_TRUNCATED = 1
E_HDR_FCHECK = 2
E_HDR_METHOD = 3
E_HDR_WINDOW = 4
E_FMT_LEN_NLEN = 5
E_FMT_BTYPE = 6
E_HUFF_BUILD = 7
E_HUFF_EMPTY_TREE = 8
E_HUFF_EMPTY_BRANCH = 9
E_CHECKSUM = 10
E_BAD_DISTANCE = 11
E_BAD_LITLEN_SYMBOL = 12
E_BAD_DIST_SYMBOL = 13
E_OUT_TOO_SMALL = 14
E_DATA_REMAINING = 15

QUIRK_REF_WINDOW_OVERFLOW = 1
QUIRK_CODELEN_OVERRUN = 2
QUIRK_REPEAT_NO_PREV = 4
QUIRK_FDICT_SKIPPED = 8
QUIRK_STOLEN_BYTE = 16
F_REF_CHUNK_BUG = 1


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("detail0", C.c_uint32),
        ("detail1", C.c_uint32),
        ("adler", C.c_uint32),
        ("out_len", C.c_uint64),
        ("in_used", C.c_uint64),
        ("quirks", C.c_uint32),
        ("n_blocks", C.c_uint32),
        ("message", C.c_char * 192),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libpzoracle.so")
    src = os.path.join(_HERE, "pz_oracle.c")
    hdr = os.path.join(_HERE, "pz_oracle.h")
    mt = os.path.join(_HERE, "pz_baseline_mt.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(mt)):
        subprocess.check_call(["make", "-C", _HERE, "libpzoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.pzo_decompress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(Result)]
        L.pzo_decompress.restype = C.c_int
        L.pzo_decompress_chunks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p,
                                            C.c_uint64, C.c_uint32, C.POINTER(Result)]
        L.pzo_decompress_chunks.restype = C.c_int
        L.pzo_adler32.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
        L.pzo_adler32.restype = C.c_uint32
        L.pzo_compute_code_values.argtypes = [C.POINTER(C.c_int)] * 2 + [C.c_int] + [C.POINTER(C.c_int)] * 3
        L.pzo_compute_code_values.restype = C.c_int
        L.pzo_decompress_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.pzo_decompress_many.restype = C.c_uint32
        _LIB = L
    return _LIB


def decompress(data: bytes, out_cap: int = None):
    """Returns (Result, output bytes truncated to min(out_len, out_cap))."""
    if out_cap is None:
        out_cap = max(1 << 16, len(data) * 1100 + 64)
    out = C.create_string_buffer(max(out_cap, 1))
    r = Result()
    lib().pzo_decompress(data, len(data), out, out_cap, C.byref(r))
    return r, out.raw[: min(r.out_len, out_cap)]


def decompress_chunks(chunks, out_cap: int = None, flags: int = 0):
    flat = b"".join(chunks)
    offs = [0]
    for c in chunks:
        offs.append(offs[-1] + len(c))
    if out_cap is None:
        out_cap = max(1 << 16, len(flat) * 1100 + 64)
    out = C.create_string_buffer(max(out_cap, 1))
    arr = (C.c_uint64 * len(offs))(*offs)
    r = Result()
    lib().pzo_decompress_chunks(flat, arr, len(chunks), out, out_cap, flags, C.byref(r))
    return r, out.raw[: min(r.out_len, out_cap)]


F_REF_CHUNK_BUG, F_GZIP = 1, 2


def gzip_decompress(data: bytes, out_cap: int = None):
    """One RFC 1952 member (extension); Result.adler holds the CRC-32."""
    return decompress_chunks([data] if data else [], out_cap, F_GZIP)


def crc32(data: bytes, init: int = 0) -> int:
    L = lib()
    L.pzo_crc32.restype = C.c_uint32
    L.pzo_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_uint64]
    return L.pzo_crc32(init, data, len(data))


def adler32(data: bytes, init: int = 1) -> int:
    return lib().pzo_adler32(init, data, len(data))


def compute_code_values(pairs):
    n = len(pairs)
    syms = (C.c_int * n)(*[p[0] for p in pairs])
    lens = (C.c_int * n)(*[p[1] for p in pairs])
    os_, ol, oc = (C.c_int * 512)(), (C.c_int * 512)(), (C.c_int * 512)()
    m = lib().pzo_compute_code_values(syms, lens, n, os_, ol, oc)
    return [(os_[i], ol[i], oc[i]) for i in range(m)]


EV_NEED_MORE, EV_CHUNK, EV_DONE, EV_ERROR = 1, 2, 3, 4


def trace(pieces, out_cap: int = None):
    """decompressIncremental driven one piece per NeedMore (Deflate.hs:30-48): the list of events
    ("NeedMore",) / ("Chunk", length) / ("Done",) / ("DecompError", status), the Result and the bytes."""
    L = lib()
    L.pzo_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                            C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(Result)]
    L.pzo_trace.restype = C.c_int
    flat = b"".join(pieces)
    offs = [0]
    for c in pieces:
        offs.append(offs[-1] + len(c))
    if out_cap is None:
        out_cap = max(1 << 16, len(flat) * 1100 + 64)
    out = C.create_string_buffer(max(out_cap, 1))
    arr = (C.c_uint64 * len(offs))(*offs)
    cap = 4 * len(pieces) + 64 + (len(flat) * 1100) // 32768
    et = (C.c_int32 * cap)()
    ev = (C.c_uint32 * cap)()
    n = C.c_uint32(0)
    r = Result()
    L.pzo_trace(flat, arr, len(pieces), out, out_cap, et, ev, cap, C.byref(n), C.byref(r))
    assert n.value <= cap
    names = {EV_NEED_MORE: "NeedMore", EV_CHUNK: "Chunk", EV_DONE: "Done", EV_ERROR: "DecompError"}
    events = [(names[et[k]],) if et[k] in (EV_NEED_MORE, EV_DONE) else (names[et[k]], int(ev[k])) for k in range(n.value)]
    return events, r, out.raw[: min(r.out_len, out_cap)]


def decompress_dict(data: bytes, zdict: bytes, out_cap: int = None):
    """EXTENSION: with the preset dictionary (RFC 1950 FDICT) installed; pinned against zlib.decompressobj(zdict=...)."""
    L = lib()
    L.pzo_decompress_dict.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(Result)]
    L.pzo_decompress_dict.restype = C.c_int
    if out_cap is None:
        out_cap = max(1 << 16, len(data) * 1100 + 64)
    out = C.create_string_buffer(max(out_cap, 1))
    r = Result()
    L.pzo_decompress_dict(data, len(data), zdict, len(zdict), out, out_cap, C.byref(r))
    return r, out.raw[: min(r.out_len, out_cap)]
