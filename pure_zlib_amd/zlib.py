"""Host-side mirror of the reference module `Codec.Compression.Zlib`
(src/Codec/Compression/Zlib.hs:3-8) over the C ABI of include/pzg.h.

    decompress      :: L.ByteString   -> Either DecompressionError L.ByteString   (Zlib.hs:32)
    decompressMany  :: [L.ByteString] -> [Either DecompressionError L.ByteString]  (new, batched)

Same names, argument meaning and error behaviour as the reference: a lazy ByteString is a
`bytes` (one chunk) or a sequence of `bytes` chunks; the result is `Right(bytes)` or
`Left(DecompressionError)` whose `show` text equals the reference's.  All decoding happens in the
HIP kernels behind libpzg.so; nothing here inflates on the CPU.
"""
import ctypes as C
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _ffi

# ---- DecompressionError (Monad.hs:87-104) -------------------------------------------------------

_SHOW_PREFIX = {
    "HuffmanTreeError": "Huffman tree manipulation error: ",
    "FormatError": "Block format error: ",
    "DecompressionError": "Decompression error: ",
    "HeaderError": "Header error: ",
    "ChecksumError": "Checksum error: ",
    # outcomes on which the reference throws a Haskell exception instead of returning Left
    # (SURVEY.md 8a a7/a16); this build reports them as a value
    "ReferenceThrows": "",
}


class DecompressionError(Exception):
    """`data DecompressionError = HuffmanTreeError String | FormatError String | DecompressionError String
    | HeaderError String | ChecksumError String  deriving (Eq)` with the reference's custom Show."""

    def __init__(self, constructor: str, message: str, status: int = -1, detail=(0, 0)):
        super().__init__(constructor, message)
        self.constructor = constructor
        self.message = message
        self.status = status
        self.detail = tuple(detail)

    def show(self) -> str:
        return _SHOW_PREFIX[self.constructor] + self.message

    __str__ = show

    def __repr__(self):
        return f"{self.constructor} {self.message!r}"

    def __eq__(self, other):
        return (isinstance(other, DecompressionError) and self.constructor == other.constructor
                and self.message == other.message)

    def __hash__(self):
        return hash((self.constructor, self.message))


def HuffmanTreeError(s):
    return DecompressionError("HuffmanTreeError", s)


def FormatError(s):
    return DecompressionError("FormatError", s)


def DecompressionError_(s):
    return DecompressionError("DecompressionError", s)


def HeaderError(s):
    return DecompressionError("HeaderError", s)


def ChecksumError(s):
    return DecompressionError("ChecksumError", s)


@dataclass(frozen=True)
class Left:
    value: DecompressionError

    def is_right(self):
        return False


@dataclass(frozen=True)
class Right:
    value: bytes

    def is_right(self):
        return True


Either = Union[Left, Right]
LazyByteString = Union[bytes, bytearray, memoryview, Sequence[bytes]]

_STATUS_CONSTRUCTOR = {
    _ffi.E_TRUNCATED: "DecompressionError",
    _ffi.E_HDR_FCHECK: "HeaderError",
    _ffi.E_HDR_METHOD: "HeaderError",
    _ffi.E_HDR_WINDOW: "HeaderError",
    _ffi.E_FMT_LEN_NLEN: "FormatError",
    _ffi.E_FMT_BTYPE: "FormatError",
    _ffi.E_HUFF_BUILD: "HuffmanTreeError",
    _ffi.E_HUFF_EMPTY_TREE: "HuffmanTreeError",
    _ffi.E_HUFF_EMPTY_BRANCH: "HuffmanTreeError",
    _ffi.E_CHECKSUM: "ChecksumError",
    _ffi.E_BAD_DISTANCE: "ReferenceThrows",
    _ffi.E_BAD_LITLEN_SYMBOL: "ReferenceThrows",
    _ffi.E_BAD_DIST_SYMBOL: "ReferenceThrows",
    _ffi.E_DATA_REMAINING: "DecompressionError",
    _ffi.E_GZIP_HEADER: "HeaderError",   # extension (PZG_GZIP): the reference has no gzip
    _ffi.E_GZIP_ISIZE: "ChecksumError",
    _ffi.E_DICT: "HeaderError",         # extension (preset dictionaries)
}


def error_from_status(stream: bytes, status: int, detail) -> DecompressionError:
    """Rebuild the reference's DecompressionError (constructor + exact message) from the ABI's
    (status, detail) pair via pzg_error_message()."""
    d = (C.c_uint32 * 2)(int(detail[0]), int(detail[1]))
    buf = C.create_string_buffer(256)
    _ffi.lib().pzg_error_message(stream, len(stream), int(status), d, buf, len(buf))
    text = buf.value.decode()
    cons = _STATUS_CONSTRUCTOR.get(int(status), "ReferenceThrows")
    prefix = _SHOW_PREFIX[cons]
    msg = text[len(prefix):] if prefix and text.startswith(prefix) else text
    return DecompressionError(cons, msg, int(status), (int(detail[0]), int(detail[1])))


# ---- context ---------------------------------------------------------------------------------------


class Context:
    """One pzg_ctx (HIP device + stream + staging arenas)."""

    def __init__(self, device: int = 0, device_mask: Optional[int] = None, devices: Optional[Sequence[int]] = None):
        """device: one HIP device.  device_mask: several devices of one node (bit d = device d, 0 = all visible):
        decompress_many then shards a host batch over them inside the library (pzg_init_mask).  devices: the same with
        the devices named one by one, one shard per entry (pzg_init_devices; a device may appear more than once)."""
        L = _ffi.lib()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            _ffi.check(L.pzg_init_devices(arr, len(devices), C.byref(h)))
        elif device_mask is None:
            _ffi.check(L.pzg_init(device, C.byref(h)))
        else:
            _ffi.check(L.pzg_init_mask(device_mask, C.byref(h)))
        self._h = h
        self._L = L
        self.device = device

    @property
    def device_count(self) -> int:
        return int(self._L.pzg_device_count(self._h))

    def close(self):
        """pzg_shutdown: drops this handle.  DecoderPools made from the context stay usable and free the device state
        when the last of them is closed (the library counts references; include/pzg.h "Lifetimes")."""
        if self._h:
            self._L.pzg_shutdown(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream: Optional[int]):
        """Launch on an existing hipStream_t (integer handle; 0/None = HIP's default stream)."""
        _ffi.check(self._L.pzg_set_stream(self._h, C.c_void_p(hip_stream or 0)), self._h)

    def reset_stream(self):
        _ffi.check(self._L.pzg_reset_stream(self._h), self._h)

    def set_ring_bits(self, ring_bits: int):
        """LDS near-ring size class (11..15; 15 = the whole 32 KiB window in LDS).  Results are identical."""
        _ffi.check(self._L.pzg_set_option(self._h, _ffi.OPT_RING_BITS, int(ring_bits)), self._h)

    def set_host_threads(self, n: int):
        """Helper threads of the staged host-pointer path and of the decoders' feeds (PZG_OPT_HOST_THREADS)."""
        _ffi.check(self._L.pzg_set_option(self._h, _ffi.OPT_HOST_THREADS, int(n)), self._h)

    def set_scratch_bytes(self, nbytes: int):
        """Upper bound, per device, on the scratch the library allocates for its inflate kernels (PZG_OPT_SCRATCH_BYTES; 0: no bound)."""
        _ffi.check(self._L.pzg_set_option(self._h, _ffi.OPT_SCRATCH_BYTES, int(nbytes)), self._h)

    def set_bundles(self, mode: int):
        """PZG_OPT_BUNDLES: device-pointer launches take their streams of the fixed code 64 to a wavefront, one lane per stream
        (0: never, 1: launches of 32,768 streams or more -- the default, 2: always).  Results are identical."""
        _ffi.check(self._L.pzg_set_option(self._h, _ffi.OPT_BUNDLES, int(mode)), self._h)

    def set_profile(self, on: bool):
        """PZG_OPT_PROFILE: the stream-waves' memory of where the last stream's tokens lay (default on).  Results are identical."""
        _ffi.check(self._L.pzg_set_option(self._h, _ffi.OPT_PROFILE, 1 if on else 0), self._h)

    def sync(self):
        _ffi.check(self._L.pzg_sync(self._h), self._h)

    def last_kernel_ms(self) -> float:
        return float(self._L.pzg_last_kernel_ms(self._h))

    # -- raw batched call on host memory ----------------------------------------------------------
    def decompress_many_raw(self, in_buf: np.ndarray, in_off, in_len, out_buf: np.ndarray, out_off, out_cap, gzip: bool = False,
                            dict_buf: Optional[np.ndarray] = None, dict_off=None, dict_len=None, pinned: bool = False):
        """Thin wrapper of pzg_decompress_many on host numpy buffers.
        pinned: in_buf / out_buf are page-locked arenas (pinned_array) with ascending extents -- PZG_HOST_PINNED.
        Returns (out_len u64[n], status i32[n], detail u32[n,2], in_used u64[n], adler u32[n])."""
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        in_len = np.ascontiguousarray(in_len, dtype=np.uint64)
        out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
        out_cap = np.ascontiguousarray(out_cap, dtype=np.uint64)
        n = len(in_off)
        out_len = np.zeros(n, dtype=np.uint64)
        status = np.full(n, -1, dtype=np.int32)
        detail = np.zeros((n, 2), dtype=np.uint32)
        in_used = np.zeros(n, dtype=np.uint64)
        adler = np.zeros(n, dtype=np.uint32)
        if n == 0:
            return out_len, status, detail, in_used, adler
        if dict_buf is not None:  # extension: one preset dictionary extent per stream (length 0 = none)
            if gzip:  # (preset dictionaries are a zlib-container notion: the C ABI rejects the combination too)
                raise ValueError("gzip=True and preset dictionaries cannot be combined")
            dict_off = np.ascontiguousarray(dict_off, dtype=np.uint64)
            dict_len = np.ascontiguousarray(dict_len, dtype=np.uint64)
            rc = self._L.pzg_decompress_many_dict(
                self._h, in_buf.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, dict_buf.ctypes.data, dict_off.ctypes.data,
                dict_len.ctypes.data, out_buf.ctypes.data, out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data,
                status.ctypes.data, detail.ctypes.data, in_used.ctypes.data, adler.ctypes.data, n, 0)
            _ffi.check(rc, self._h)
            return out_len, status, detail, in_used, adler
        rc = self._L.pzg_decompress_many(
            self._h, in_buf.ctypes.data, in_off.ctypes.data, in_len.ctypes.data, out_buf.ctypes.data,
            out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data, status.ctypes.data, detail.ctypes.data,
            in_used.ctypes.data, adler.ctypes.data, n, (_ffi.GZIP if gzip else 0) | (_ffi.HOST_PINNED if pinned else 0))
        _ffi.check(rc, self._h)
        return out_len, status, detail, in_used, adler

    # -- device-pointer call (the timed path; pointers are raw integers) ----------------------------
    def decompress_many_device(self, in_base: int, in_off: int, in_len: int, out_base: int, out_off: int,
                               out_cap: int, out_len: int, status: int, detail: int, in_used: int, adler: int,
                               n: int, sync: bool = True, gzip: bool = False, lpt: bool = False):
        flags = _ffi.DEVICE_PTRS | (0 if sync else _ffi.ASYNC) | (_ffi.GZIP if gzip else 0) | (_ffi.LPT_ORDER if lpt else 0)
        rc = self._L.pzg_decompress_many(self._h, in_base, in_off, in_len, out_base, out_off, out_cap, out_len,
                                         status, detail or None, in_used or None, adler or None, n, flags)
        _ffi.check(rc, self._h)

    def decompress_many_sharded(self, batches, sync: bool = True, gzip: bool = False, lpt: bool = False):
        """pzg_decompress_many_sharded: `batches` is a list of dicts (shard, n, in_base, in_off, in_len, out_base, out_off,
        out_cap, out_len, status and optionally detail, in_used, adler), every pointer an integer address of device memory
        on that shard's device.  One call enqueues them all; nothing leaves the devices."""
        arr = (_ffi.DeviceBatch * len(batches))()
        for q, b in zip(arr, batches):
            for k, _t in _ffi.DeviceBatch._fields_:
                setattr(q, k, b.get(k) or 0)
        flags = (0 if sync else _ffi.ASYNC) | (_ffi.GZIP if gzip else 0) | (_ffi.LPT_ORDER if lpt else 0)
        _ffi.check(self._L.pzg_decompress_many_sharded(self._h, C.byref(arr) if len(batches) else None, len(batches), flags), self._h)

    def adler32(self, data, init: int = 1) -> int:
        arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = C.c_uint32(0)
        ptr = arr.ctypes.data if arr.size else None
        _ffi.check(self._L.pzg_adler32(self._h, ptr, arr.size, init, C.byref(out), 0), self._h)
        return out.value

    def adler32_many_device(self, base: int, off: int, length: int, out_ptr: int, n: int, sync: bool = True):
        """Adler-32 of n device buffers (one wave each): BASELINE config 2 in its batched form."""
        flags = _ffi.DEVICE_PTRS | (0 if sync else _ffi.ASYNC)
        _ffi.check(self._L.pzg_adler32_many(self._h, base, off, length, out_ptr, n, flags), self._h)

    def adler32_device(self, ptr: int, nbytes: int, out_ptr: int, init: int = 1, sync: bool = True):
        flags = _ffi.DEVICE_PTRS | (0 if sync else _ffi.ASYNC)
        _ffi.check(self._L.pzg_adler32(self._h, ptr, nbytes, init, out_ptr, flags), self._h)


class PinnedArena:
    """A page-locked host buffer from pzg_host_alloc, seen as a numpy uint8 array (`.a`).  The copy engines read and write
    it directly (PZG_HOST_PINNED): what the mirrors pack their batches into, so that nothing is staged a second time."""

    def __init__(self, nbytes: int):
        self._L = _ffi.lib()
        self.nbytes = max(int(nbytes), 1)
        self._p = self._L.pzg_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError(f"pzg_host_alloc({self.nbytes}) failed: the system will not lock that much memory")
        self.a = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self._p))

    def close(self):
        if self._p:
            self.a = None
            self._L.pzg_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# the mirrors' arenas: grow-only, one pair per thread (a `decompress` call is synchronous)
_arenas = threading.local()


def _arena(kind: str, nbytes: int) -> Optional[np.ndarray]:
    """A page-locked array of at least nbytes for this thread's `kind` ("in" / "out"), or None when the system refuses."""
    cur = getattr(_arenas, kind, None)
    if cur is None or cur.nbytes < nbytes:
        if cur is not None:
            cur.close()
        try:
            cur = PinnedArena(max(nbytes, 1 << 20) * 5 // 4)
        except MemoryError:
            setattr(_arenas, kind, None)
            return None
        setattr(_arenas, kind, cur)
    return cur.a


_default_ctx = None
_default_lock = threading.Lock()


def default_context() -> Context:
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context(0)
        return _default_ctx


# ---- the reference API -------------------------------------------------------------------------------


def _to_chunks(x: LazyByteString) -> List[bytes]:
    if isinstance(x, (bytes, bytearray, memoryview)):
        b = bytes(x)
        return [b] if b else []
    return [bytes(c) for c in x if len(c)]  # a lazy ByteString never holds empty chunks


def _align(x, a=256):
    return (x + a - 1) // a * a


def decompress_many(streams: Sequence[LazyByteString], ctx: Optional[Context] = None,
                    size_hint: Optional[Sequence[int]] = None, gzip: bool = False,
                    zdict: Optional[Sequence[Optional[bytes]]] = None) -> List[Either]:
    """decompressMany: every stream decoded by its own wavefront in one launch.

    zlib streams do not carry their decoded size, so each stream gets a capacity (size_hint[i] or a
    guess); streams that report PZG_E_OUT_TOO_SMALL are relaunched once with the exact size the
    kernel measured.

    zdict (EXTENSION, the reference skips DICTID: Zlib.hs:68): one preset dictionary per stream (None = none); a stream
    whose header has FDICT set then decodes with it as history, as zlib.decompressobj(zdict=...) does."""
    ctx = ctx or default_context()
    chunked = [_to_chunks(s) for s in streams]
    flat = [b"".join(c) for c in chunked]
    n = len(flat)
    results: List[Optional[Either]] = [None] * n
    # without a hint: a modest first guess (the batch is packed, so small guesses cost little); PZG_E_OUT_TOO_SMALL
    # streams come back with their exact size and are relaunched once
    caps = [int(size_hint[i]) if size_hint is not None else max(256, 4 * len(flat[i])) for i in range(n)]
    todo = list(range(n))
    for _attempt in range(2):
        if not todo:
            break
        m = len(todo)
        in_off = np.zeros(m, dtype=np.uint64)
        in_len = np.array([len(flat[i]) for i in todo], dtype=np.uint64)
        out_off = np.zeros(m, dtype=np.uint64)
        out_cap = np.array([caps[i] for i in todo], dtype=np.uint64)
        ipos = opos = 0
        for k in range(m):  # 16-byte aligned extents: the wide store path
            in_off[k], out_off[k] = ipos, opos
            ipos += _align(int(in_len[k]), 16)
            opos += _align(int(out_cap[k]), 16)
        # packed into page-locked arenas (PZG_HOST_PINNED: the library then stages nothing a second time); preset dictionaries
        # and a system that will not lock the memory take the staged path over ordinary arrays
        in_buf = out_buf = None
        if zdict is None:
            in_buf, out_buf = _arena("in", ipos + 16), _arena("out", opos + 16)
        pinned = in_buf is not None and out_buf is not None
        if not pinned:
            in_buf = np.zeros(ipos + 16, dtype=np.uint8)
            out_buf = np.zeros(opos + 16, dtype=np.uint8)
        for k, i in enumerate(todo):
            in_buf[int(in_off[k]):int(in_off[k]) + len(flat[i])] = np.frombuffer(flat[i], dtype=np.uint8)
        dict_args = {}
        if zdict is not None:
            dl = np.array([len(zdict[i] or b"") for i in todo], dtype=np.uint64)
            do = np.zeros(m, dtype=np.uint64)
            do[1:] = np.cumsum(dl[:-1])
            db = np.frombuffer(b"".join((zdict[i] or b"") for i in todo) + b"\0" * 16, dtype=np.uint8)
            dict_args = dict(dict_buf=db, dict_off=do, dict_len=dl)
        out_len, status, detail, in_used, _adler = ctx.decompress_many_raw(in_buf, in_off, in_len, out_buf, out_off, out_cap, gzip,
                                                                           pinned=pinned, **dict_args)
        retry = []
        for k, i in enumerate(todo):
            st = int(status[k])
            if st == _ffi.OK:
                data = out_buf[int(out_off[k]):int(out_off[k]) + int(out_len[k])].tobytes()
                results[i] = _apply_chunk_rule(chunked[i], int(in_used[k]), data)
            elif st == _ffi.E_OUT_TOO_SMALL and _attempt == 0:
                caps[i] = int(out_len[k])
                retry.append(i)
            else:
                results[i] = Left(error_from_status(flat[i], st, detail[k]))
        todo = retry
    return results  # type: ignore[return-value]


def _apply_chunk_rule(chunks: List[bytes], in_used: int, data: bytes) -> Either:
    """Zlib.hs:46-49: `Done` with whole unread chunks left is `Left "Finished with data remaining."`;
    trailing bytes inside the last chunk handed over are silently ignored."""
    cum = 0
    loaded = 0
    for c in chunks:
        if cum >= in_used:
            break
        cum += len(c)
        loaded += 1
    if loaded < len(chunks):
        return Left(DecompressionError("DecompressionError", "Finished with data remaining.", _ffi.E_DATA_REMAINING))
    return Right(data)


def decompress(ifile: LazyByteString, ctx: Optional[Context] = None, size_hint: Optional[int] = None) -> Either:
    """Codec.Compression.Zlib.decompress (Zlib.hs:32-51)."""
    return decompress_many([ifile], ctx, None if size_hint is None else [size_hint])[0]


def adler32(data: bytes, init: int = 1, ctx: Optional[Context] = None) -> int:
    """Codec.Compression.Zlib.Adler32 over one buffer, as a device-wide reduction (Adler32.hs:17-57)."""
    return (ctx or default_context()).adler32(data, init)


def gzip_decompress_many(streams: Sequence[LazyByteString], ctx: Optional[Context] = None,
                         size_hint: Optional[Sequence[int]] = None) -> List[Either]:
    """EXTENSION (the reference lists gzip as a TODO): decompress_many over RFC 1952
    members -- same DEFLATE kernel, gzip header, CRC-32 + ISIZE trailer verified on the device."""
    return decompress_many(streams, ctx, size_hint, gzip=True)


# Haskell-cased aliases so call sites read like the reference
decompressMany = decompress_many
