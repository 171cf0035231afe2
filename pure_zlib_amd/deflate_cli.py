"""The reference's `deflate` executable (/Deflate.hs:15-48) over the GPU path:  deflate foo.z -> foo.

SURVEY.md section 8f row 2 ("next" row).  Same messages as the reference; the file is read in the chunk
size `L.readFile` uses (bytestring's defaultChunkSize = 32 KiB minus two words) and driven through
the ZlibDecoder protocol exactly like `runDecompression` (Deflate.hs:30-48).
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


def main(argv=None) -> int:
    args = sys.argv[1:] if argv is None else argv
    if len(args) != 1:
        print("USAGE: deflate [filename]")
        return 0
    ifile = args[0]
    if not ifile.endswith(".z"):
        print("Unexpected file name.")
        return 0
    with open(ifile, "rb") as f:
        data = f.read()
    chunks = [data[i:i + LAZY_CHUNK] for i in range(0, len(data), LAZY_CHUNK)]
    with open(ifile[:-2], "wb") as out:
        run_decompression(out, chunks, decompress_incremental())
    return 0


if __name__ == "__main__":
    sys.exit(main())
