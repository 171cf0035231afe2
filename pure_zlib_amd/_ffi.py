"""ctypes binding of libpzg.so (include/pzg.h).

The library is the only compute path of this package: if it is missing, or HIP has no usable
device, the calls raise.  There is no CPU fallback anywhere in pure_zlib_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PZG_LIB") or os.path.join(_HERE, "libpzg.so")  # PZG_LIB: a diagnostic build (tests/tools/exp_build.sh)

RC_OK, RC_BAD_ARG, RC_NO_DEVICE, RC_HIP_ERROR, RC_NO_MEMORY = 0, -1, -2, -3, -4

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
E_GZIP_HEADER = 18  # PZG_GZIP only (extension)
E_GZIP_ISIZE = 19
E_DICT = 20  # pzg_decompress_many_dict only (extension)
DEC_NEED_INPUT = 101  # pzg_decoder_feed: NeedMore
DEC_OUT_FULL = 102    # pzg_decoder_feed: this call's output room is used up

DEVICE_PTRS = 1
ASYNC = 2
GZIP = 4  # extension: streams are RFC 1952 members; adler[] holds the CRC-32
LPT_ORDER = 8  # device-pointer batches: launch the longest streams first
HOST_PINNED = 16  # host-pointer batches in page-locked arenas (pzg_host_alloc), extents ascending: no staging, no copy-out
OPT_RING_BITS = 1
OPT_HOST_THREADS = 2
OPT_SCRATCH_BYTES = 3
OPT_BUNDLES = 4
OPT_PROFILE = 5
DEFAULT_RING_BITS = 11

# every symbol include/pzg.h declares
SYMBOLS = [
    "pzg_init", "pzg_init_mask", "pzg_init_devices", "pzg_host_alloc", "pzg_host_free", "pzg_device_count", "pzg_adler32_many", "pzg_decompress_many_dict", "pzg_decompress_many_sharded",
    "pzg_decoder_create", "pzg_decoder_destroy", "pzg_decoder_reset", "pzg_decoder_feed", "pzg_decoder_last_feed_ms", "pzg_shutdown", "pzg_set_stream", "pzg_reset_stream", "pzg_set_option", "pzg_sync", "pzg_decompress_many", "pzg_decompress",
    "pzg_adler32", "pzg_error_message", "pzg_last_kernel_ms", "pzg_strerror", "pzg_last_error", "pzg_version",
]


class DeviceBatch(C.Structure):
    """pzg_device_batch (include/pzg.h): one shard's share of a pzg_decompress_many_sharded call, device pointers as integers."""
    _fields_ = [("shard", C.c_uint32), ("n", C.c_uint32),
                ("in_base", C.c_void_p), ("in_off", C.c_void_p), ("in_len", C.c_void_p),
                ("out_base", C.c_void_p), ("out_off", C.c_void_p), ("out_cap", C.c_void_p),
                ("out_len", C.c_void_p), ("status", C.c_void_p), ("detail", C.c_void_p), ("in_used", C.c_void_p), ("adler", C.c_void_p)]


class PzgError(RuntimeError):
    """Call-level failure of the library (not a per-stream DecompressionError)."""


_lib = None


def lib():
    """Load libpzg.so.  Raises if the HIP extension has not been built: the product has no other path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PzgError(
            f"{LIB_PATH} is missing: build it with `make -C pure_zlib_amd/csrc` "
            "(or __graft_entry__.build()).  pure_zlib_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u64p, i32p, u32p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.pzg_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.pzg_init.restype = C.c_int
    L.pzg_init_mask.argtypes = [C.c_uint32, C.POINTER(C.c_void_p)]
    L.pzg_init_mask.restype = C.c_int
    L.pzg_init_devices.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.pzg_init_devices.restype = C.c_int
    L.pzg_host_alloc.argtypes = [C.c_size_t]
    L.pzg_host_alloc.restype = C.c_void_p
    L.pzg_host_free.argtypes = [C.c_void_p]
    L.pzg_host_free.restype = None
    L.pzg_device_count.argtypes = [C.c_void_p]
    L.pzg_device_count.restype = C.c_int
    L.pzg_adler32_many.argtypes = [C.c_void_p, vp, u64p, u64p, u32p, C.c_uint32, C.c_uint32]
    L.pzg_adler32_many.restype = C.c_int
    L.pzg_shutdown.argtypes = [C.c_void_p]
    L.pzg_shutdown.restype = None
    L.pzg_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.pzg_set_stream.restype = C.c_int
    L.pzg_reset_stream.argtypes = [C.c_void_p]
    L.pzg_reset_stream.restype = C.c_int
    L.pzg_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    L.pzg_set_option.restype = C.c_int
    L.pzg_sync.argtypes = [C.c_void_p]
    L.pzg_sync.restype = C.c_int
    L.pzg_decompress_many.argtypes = [C.c_void_p, vp, u64p, u64p, vp, u64p, u64p, u64p, i32p, u32p, u64p, u32p,
                                      C.c_uint32, C.c_uint32]
    L.pzg_decompress_many.restype = C.c_int
    L.pzg_decompress_many_dict.argtypes = [C.c_void_p, vp, u64p, u64p, vp, u64p, u64p, vp, u64p, u64p, u64p, i32p, u32p, u64p, u32p,
                                           C.c_uint32, C.c_uint32]
    L.pzg_decompress_many_dict.restype = C.c_int
    L.pzg_decompress_many_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.pzg_decompress_many_sharded.restype = C.c_int
    L.pzg_decoder_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.pzg_decoder_create.restype = C.c_int
    L.pzg_decoder_destroy.argtypes = [C.c_void_p]
    L.pzg_decoder_destroy.restype = None
    L.pzg_decoder_reset.argtypes = [C.c_void_p, u32p, C.c_uint32]
    L.pzg_decoder_reset.restype = C.c_int
    L.pzg_decoder_feed.argtypes = [C.c_void_p, u32p, C.c_uint32, vp, u64p, u64p, vp, vp, u64p, u64p, u64p, i32p, u32p, u64p, u32p, u32p]
    L.pzg_decoder_feed.restype = C.c_int
    L.pzg_decoder_last_feed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.pzg_decoder_last_feed_ms.restype = C.c_int
    L.pzg_decompress.argtypes = [C.c_void_p, vp, C.c_uint64, vp, C.c_uint64, u64p, i32p, u32p, u64p]
    L.pzg_decompress.restype = C.c_int
    L.pzg_adler32.argtypes = [C.c_void_p, vp, C.c_uint64, C.c_uint32, u32p, C.c_uint32]
    L.pzg_adler32.restype = C.c_int
    L.pzg_error_message.argtypes = [vp, C.c_uint64, C.c_int32, u32p, C.c_char_p, C.c_size_t]
    L.pzg_error_message.restype = C.c_int
    L.pzg_last_kernel_ms.argtypes = [C.c_void_p]
    L.pzg_last_kernel_ms.restype = C.c_double
    L.pzg_strerror.argtypes = [C.c_int]
    L.pzg_strerror.restype = C.c_char_p
    L.pzg_last_error.argtypes = [C.c_void_p]
    L.pzg_last_error.restype = C.c_char_p
    L.pzg_version.argtypes = []
    L.pzg_version.restype = C.c_uint32
    _lib = L
    return L


def check(rc, ctx=None):
    if rc != RC_OK:
        L = lib()
        msg = L.pzg_strerror(rc).decode()
        if ctx is not None and rc == RC_HIP_ERROR:
            msg += ": " + L.pzg_last_error(ctx).decode()
        raise PzgError(f"libpzg call failed ({rc}): {msg}")
