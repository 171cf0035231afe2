"""`decompressIncremental :: ST s (ZlibDecoder s)` (src/Codec/Compression/Zlib.hs:29-30) and the
`ZlibDecoder` protocol (src/Codec/Compression/Zlib/Monad.hs:163-167) over resumable decoders on the GPU.

    data ZlibDecoder s = NeedMore (ByteString -> ST s (ZlibDecoder s)) | Chunk ByteString (ST s (ZlibDecoder s))
                       | Done | DecompError DecompressionError

The reference's decoder is a resumable CPS computation: it suspends wherever the input runs out and publishes
32,768-byte chunks as soon as 64 KiB are buffered (moveWindow after every match and at every block end,
Monad.hs:338-347, OutputWindow.hs:45-54), interleaved with NeedMore.  Here the suspended decoder lives in HBM
(`pzg_decoder`, include/pzg.h): a feed is ONE launch that continues it from where it stopped -- nothing is
re-decoded -- and reports how many chunks the reference has published by then, so the constructors below come out
in exactly the reference's order (tests compare the event trace with the oracle's).  A `DecoderPool` continues many
decoders per launch; `decompress_incremental()` is the reference's single-decoder entry point on a pool of one.
Nothing here inflates on the CPU.
"""
import functools
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .zlib import Context, DecompressionError, default_context, error_from_status

EXCESS_CHUNK = 32768  # OutputWindow.hs:42-43 excessChunkSize
_ROOM = 256 * 1024    # output room per decoder and launch


class Done:
    def __repr__(self):
        return "Done"


class DecompError:
    def __init__(self, error: DecompressionError):
        self.error = error

    def __repr__(self):
        return f"DecompError ({self.error!r})"


class Chunk:
    def __init__(self, chunk: bytes, rest):
        self.chunk = chunk
        self._rest = rest

    def next(self):
        """Run the continuation (the `ST s (ZlibDecoder s)` of the constructor)."""
        return self._rest()

    def __repr__(self):
        return f"Chunk <{len(self.chunk)} bytes>"


class NeedMore:
    def __init__(self, pool: "DecoderPool", k: int):
        self._pool, self._k = pool, k

    def feed(self, chunk: bytes):
        """Apply the continuation to the next input chunk (Monad.hs:185-197 loadChunk)."""
        return self._pool.feed([self._k], [chunk])[0]

    def __repr__(self):
        return "NeedMore"


class DecoderPool:
    """n resumable zlib decoders on the device (pzg_decoder_create); feed() continues any subset of them in one launch."""

    def __init__(self, n: int, ctx: Optional[Context] = None, room: int = _ROOM):
        self._ctx = ctx or default_context()
        self._L = _ffi.lib()
        import ctypes as C
        h = C.c_void_p()
        _ffi.check(self._L.pzg_decoder_create(self._ctx.handle, n, C.byref(h)), self._ctx.handle)
        self._h = h
        self.n = n
        self._room = max(4096, int(room))
        self._tail: List[bytes] = [b""] * n        # input the decoder has not consumed yet
        self._pending = [bytearray() for _ in range(n)]  # delivered bytes not yet published as chunks
        self._published = [0] * n                  # chunks published so far
        self._all = [bytearray() for _ in range(n)]  # (kept for error_from_status: the whole input so far)
        self._closed = [False] * n

    def close(self):
        """pzg_decoder_destroy.  Legal before or after the Context is closed: the decoders hold their own reference on
        the library's context (include/pzg.h "Lifetimes"), as a ZlibDecoder closure keeps what it needs alive."""
        if self._h:
            self._L.pzg_decoder_destroy(self._h)
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

    def start(self, k: int) -> NeedMore:
        """`runDeflateM` starts with no input (Monad.hs:172-179): the first state is always NeedMore."""
        return NeedMore(self, k)

    def feed(self, ks: Sequence[int], chunks: Sequence[bytes], final: bool = False):
        """Continue decoders ks[j] with chunks[j]; returns their next constructors (a chain of Chunks ending in
        NeedMore / Done / DecompError)."""
        todo = []
        results = {}
        for k, c in zip(ks, chunks):
            if self._closed[k]:
                raise ValueError("this decoder has finished")
            if not self._h:
                raise ValueError("this DecoderPool has been closed")
            if len(c) == 0 and not final:
                results[k] = NeedMore(self, k)  # S.uncons = Nothing: ask again (Monad.hs:194)
            else:
                self._tail[k] += bytes(c)
                self._all[k] += bytes(c)
                todo.append(k)
        events = {k: [] for k in todo}
        active = list(todo)
        while active:  # (a decoder that runs out of room is continued with the rest of its input)
            m = len(active)
            idx = np.array(active, dtype=np.uint32)
            in_len = np.array([len(self._tail[k]) for k in active], dtype=np.uint64)
            in_off = np.zeros(m, dtype=np.uint64)
            in_off[1:] = np.cumsum(in_len[:-1])
            in_buf = np.frombuffer(b"".join(self._tail[k] for k in active) + b"\0" * 16, dtype=np.uint8)
            out_cap = np.full(m, self._room, dtype=np.uint64)
            out_off = np.arange(m, dtype=np.uint64) * np.uint64(self._room)
            out_buf = np.zeros(m * self._room + 16, dtype=np.uint8)
            out_len = np.zeros(m, dtype=np.uint64)
            state = np.zeros(m, dtype=np.int32)
            detail = np.zeros((m, 2), dtype=np.uint32)
            in_used = np.zeros(m, dtype=np.uint64)
            chunks = np.zeros(m, dtype=np.uint32)
            fin = np.full(m, 1 if final else 0, dtype=np.uint8)
            rc = self._L.pzg_decoder_feed(self._h, idx.ctypes.data, m, in_buf.ctypes.data, in_off.ctypes.data, in_len.ctypes.data,
                                          fin.ctypes.data, out_buf.ctypes.data, out_off.ctypes.data, out_cap.ctypes.data,
                                          out_len.ctypes.data, state.ctypes.data, detail.ctypes.data, in_used.ctypes.data,
                                          chunks.ctypes.data, None)
            _ffi.check(rc, self._ctx.handle)
            again = []
            for j, k in enumerate(active):
                self._pending[k] += out_buf[int(out_off[j]):int(out_off[j]) + int(out_len[j])].tobytes()
                self._tail[k] = self._tail[k][int(in_used[j]):]
                while self._published[k] < int(chunks[j]):  # what moveWindow has published by now
                    events[k].append(("Chunk", bytes(self._pending[k][:EXCESS_CHUNK])))
                    del self._pending[k][:EXCESS_CHUNK]
                    self._published[k] += 1
                st = int(state[j])
                if st == _ffi.DEC_OUT_FULL:
                    again.append(k)
                elif st == _ffi.DEC_NEED_INPUT:
                    events[k].append(("NeedMore",))
                elif st == _ffi.OK:
                    events[k].append(("Chunk", bytes(self._pending[k])))  # finalize (Monad.hs:349-353): the rest, as one chunk
                    self._pending[k] = bytearray()
                    events[k].append(("Done",))
                    self._closed[k] = True
                else:
                    events[k].append(("DecompError", error_from_status(bytes(self._all[k]), st, detail[j])))
                    self._closed[k] = True
            active = again
        for k in todo:
            results[k] = self._chain(k, events[k])
        return [results[k] for k in ks]

    def _chain(self, k, events):
        return _state_at(self, k, events, 0)


def _state_at(pool: "DecoderPool", k: int, events, i: int):
    """The constructor for events[i]; a Chunk's continuation is a partial of this module-level function (no closure
    that refers to itself: a dropped decoder is freed by reference counting at once, not by a later cycle collection)."""
    e = events[i]
    if e[0] == "Chunk":
        return Chunk(e[1], functools.partial(_state_at, pool, k, events, i + 1))
    if e[0] == "NeedMore":
        return NeedMore(pool, k)
    if e[0] == "Done":
        return Done()
    return DecompError(e[1])


def decompress_incremental(ctx: Optional[Context] = None):
    """decompressIncremental: the initial decoder state (always NeedMore)."""
    return DecoderPool(1, ctx).start(0)


decompressIncremental = decompress_incremental
