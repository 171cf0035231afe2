"""`decompressIncremental :: ST s (ZlibDecoder s)` (src/Codec/Compression/Zlib.hs:29-30) and the
`ZlibDecoder` protocol (src/Codec/Compression/Zlib/Monad.hs:163-167) over the batched GPU path.

    data ZlibDecoder s = NeedMore (ByteString -> ST s (ZlibDecoder s)) | Chunk ByteString (ST s (ZlibDecoder s))
                       | Done | DecompError DecompressionError

SURVEY.md section 8f row 1 ("next" row): the reference's decoder is a resumable CPS computation on the
CPU; a wavefront cannot be suspended mid-stream, so this mirror buffers the chunks it is fed and
re-decodes the accumulated input on the GPU each time (no CPU inflate anywhere).  It yields `NeedMore`
while the stream is incomplete and then the output as 32,768-byte `Chunk`s followed by the remainder,
the sizes `moveWindow`/`finalize` produce (Monad.hs:338-358, OutputWindow.hs:45-60).  Known differences,
by construction: the reference can hand out early chunks before it has seen the end of the input,
this mirror hands all of them out once the stream is complete; feeding n chunks costs n launches.
"""
from typing import Optional

from . import _ffi
from .zlib import Context, DecompressionError, Left, decompress

EXCESS_CHUNK = 32768  # OutputWindow.hs:42-43 excessChunkSize


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
    def __init__(self, acc: bytes, ctx: Optional[Context]):
        self._acc = acc
        self._ctx = ctx

    def feed(self, chunk: bytes):
        """Apply the continuation to the next input chunk (Monad.hs:185-197 loadChunk)."""
        if len(chunk) == 0:
            return NeedMore(self._acc, self._ctx)  # S.uncons = Nothing: ask again
        acc = self._acc + bytes(chunk)
        res = decompress(acc, ctx=self._ctx)
        if isinstance(res, Left):
            if res.value.status == _ffi.E_TRUNCATED:
                return NeedMore(acc, self._ctx)
            return DecompError(res.value)
        out = res.value
        pieces = []
        pos = 0
        while len(out) - pos >= 2 * EXCESS_CHUNK:  # emitExcess: a 32 KiB piece whenever >= 64 KiB are buffered
            pieces.append(out[pos:pos + EXCESS_CHUNK])
            pos += EXCESS_CHUNK
        pieces.append(out[pos:])  # finalizeWindow publishes whatever is left (possibly empty)

        def make(i):
            if i == len(pieces):
                return Done()
            return Chunk(pieces[i], lambda: make(i + 1))
        return make(0)

    def __repr__(self):
        return "NeedMore"


def decompress_incremental(ctx: Optional[Context] = None):
    """decompressIncremental: the initial decoder state.  `runDeflateM` starts with no input
    (Monad.hs:172-179), so the first state is always NeedMore."""
    return NeedMore(b"", ctx)


decompressIncremental = decompress_incremental
