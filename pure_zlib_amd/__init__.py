"""pure_zlib_amd -- MI355X-native batched zlib/DEFLATE decompression behind pure-zlib's
`Codec.Compression.Zlib.decompress` API.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels (gfx950) + the C ABI of include/pzg.h  -> libpzg.so
  zlib.py    host-side mirror of the reference module Codec.Compression.Zlib
  shard.py   host-side sharding of a batch of streams over the GPUs of a node
  incremental.py, deflate_cli.py, benchmark.py   the reference's streaming protocol, CLI and criterion harness
             over the same path; gzip members (row 4) are `gzip_decompress_many`
  cxx/       the same module mirror in C++ (header-only)

There is no CPU fallback: importing works anywhere, computing needs libpzg.so and a gfx950 device.
"""
from . import _ffi  # noqa: F401
from .zlib import (  # noqa: F401
    ChecksumError, Context, DecompressionError, DecompressionError_, FormatError, HeaderError,
    HuffmanTreeError, Left, Right, adler32, decompress, decompress_many, decompressMany, default_context,
    gzip_decompress_many,
)
