"""The reference's `deflate` executable (/Deflate.hs:15-48) over the GPU path:  deflate foo.z -> foo.

SURVEY.md section 8f row 2 ("next" row).  Same messages as the reference; the file is read in the chunk
size `L.readFile` uses (bytestring's defaultChunkSize = 32 KiB minus two words) and driven through
the ZlibDecoder protocol exactly like `runDecompression` (Deflate.hs:30-48).

Batch mode (on top of the reference; behind its own flag so that everything the reference's
binary prints -- "USAGE: deflate [filename]" for anything but exactly one argument -- stays as it is):
deflate --many a.z b.z c.z ...  decodes every file in ONE `decompressMany` call -- one wavefront per file, one launch -- and writes a, b, c.  Each file is handed over as the lazy
ByteString `L.readFile` would make of it, so `decompress`'s own outcomes apply per file: "ERROR: <show e>" for a Left
(including Zlib.hs:48-49's "Finished with data remaining."), "Unexpected file name." for a name that does not end in
".z"; the other files are still decoded.
"""
import sys

from .incremental import Chunk, DecompError, Done, NeedMore, decompress_incremental

LAZY_CHUNK = 32 * 1024 - 16


def run_decompression(out, chunks, decoder) -> None:
    while True:
        if isinstance(decoder, Done):
            if chunks:
                print("WARNING: Finished decompression with data left.")
            return
        if isinstance(decoder, DecompError):
            print("ERROR: " + decoder.error.show())
            return
        if isinstance(decoder, NeedMore):
            if chunks:
                decoder = decoder.feed(chunks.pop(0))
                continue
            print("ERROR: Ran out of data mid-decompression.")
            return
        if isinstance(decoder, Chunk):
            out.write(decoder.chunk)
            decoder = decoder.next()


def _lazy_chunks(data: bytes):
    return [data[i:i + LAZY_CHUNK] for i in range(0, len(data), LAZY_CHUNK)]


def run_many(files) -> None:
    """Batch mode: every `.z` file of `files` through one decompress_many call."""
    from .zlib import decompress_many
    good = []
    for f in files:
        if f.endswith(".z"):
            good.append(f)
        else:
            print(f"{f}: Unexpected file name.")
    streams = []
    for f in good:
        with open(f, "rb") as h:
            streams.append(_lazy_chunks(h.read()))
    for f, r in zip(good, decompress_many(streams)):
        if r.is_right():
            with open(f[:-2], "wb") as out:
                out.write(r.value)
        else:
            print(f"{f}: ERROR: " + r.value.show())


def main(argv=None) -> int:
    args = sys.argv[1:] if argv is None else argv
    if args and args[0] == "--many":
        run_many(args[1:])
        return 0
    if len(args) != 1:  # Deflate.hs:17-29
        print("USAGE: deflate [filename]")
        return 0
    ifile = args[0]
    if not ifile.endswith(".z"):
        print("Unexpected file name.")
        return 0
    with open(ifile, "rb") as f:
        data = f.read()
    chunks = _lazy_chunks(data)
    with open(ifile[:-2], "wb") as out:
        run_decompression(out, chunks, decompress_incremental())
    return 0


if __name__ == "__main__":
    sys.exit(main())
