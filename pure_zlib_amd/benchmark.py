"""The reference's criterion harness (Benchmark.hs:12-87, pure-zlib.cabal:84-100) over this build.

SURVEY.md section 8f row 3.  Same grouping as the reference:

    decompression/<case>/normal/{pzgpu, zlib}          Benchmark.hs:37-40
    decompression/<case>/incremental/{pzgpu, zlib}     Benchmark.hs:41-45, drivers at :53-87

with two changes the survey asks for: every sample forces the FULL output and compares it with the
`.gold` file (the reference benches `whnf`, which stops at the first lazy chunk), and a third group
sweeps the batch size of `decompressMany`:

    batch/<case>/n=<streams>/pzgpu                     one launch, every output checked

`pzgpu` is this library (host buffers in, host buffers out: H2D + kernel + D2H, the cost a
`decompress` caller pays); `zlib` is system zlib through Python, standing in for the `zlib` package
the reference compares itself with.  Cases are `<name>.z`/`<name>.gold` pairs in a directory
(default: tests/golden/ref, the reference's own nine fixtures).

    python -m pure_zlib_amd.benchmark [--dir D] [--cases a,b] [--time-limit S] [--batch 1,64,4096] [--list]
"""
import argparse
import os
import statistics
import sys
import time
import zlib as _czlib
from typing import Callable, Dict, List, Optional, Tuple

DEFAULT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref")
READ_CHUNK = 32768  # L.readFile hands out defaultChunkSize pieces


def find_cases(directory: str) -> List[str]:
    """Names with both <name>.z and <name>.gold present, sorted (Benchmark.hs:12-24 lists them by hand)."""
    names = sorted(f[:-2] for f in os.listdir(directory) if f.endswith(".z"))
    return [n for n in names if os.path.exists(os.path.join(directory, n + ".gold"))]


def get_files(directory: str, tc: str) -> Tuple[bytes, bytes]:
    """Benchmark.hs:47-50 getFiles."""
    with open(os.path.join(directory, tc + ".z"), "rb") as f:
        z = f.read()
    with open(os.path.join(directory, tc + ".gold"), "rb") as f:
        gold = f.read()
    return z, gold


def lazy_chunks(b: bytes) -> List[bytes]:
    return [b[i:i + READ_CHUNK] for i in range(0, len(b), READ_CHUNK)] or [b""]


def measure(fn: Callable[[], None], time_limit: float, min_samples: int = 3) -> Dict[str, float]:
    """A small stand-in for criterion's sampling: one untimed call, then samples until the time limit."""
    fn()
    samples: List[float] = []
    t_end = time.perf_counter() + time_limit
    while len(samples) < min_samples or time.perf_counter() < t_end:
        t0 = time.perf_counter()
        fn()
        samples.append(time.perf_counter() - t0)
        if len(samples) >= 10000:
            break
    return {"mean": statistics.fmean(samples), "stddev": statistics.pstdev(samples), "min": min(samples),
            "samples": float(len(samples))}


def fmt_time(s: float) -> str:
    for unit, k in (("s", 1.0), ("ms", 1e-3), ("us", 1e-6), ("ns", 1e-9)):
        if s >= k:
            return f"{s / k:8.3f} {unit}"
    return f"{s / 1e-9:8.3f} ns"


def build_benchmarks(directory: str, cases: List[str], batch_sizes: List[int], ctx=None):
    """Returns [(name, thunk, output_bytes_per_call)] in the reference's order."""
    from . import zlib as pz
    from .incremental import Chunk, DecompError, Done, NeedMore, decompress_incremental
    ctx = ctx or pz.default_context()
    out = []

    def incremental_pzgpu(chunks: List[bytes], gold: bytes):
        # Benchmark.hs:53-70 decompressIncrementalPure
        def run():
            st = decompress_incremental(ctx)
            rest = list(chunks)
            got = []
            while True:
                if isinstance(st, NeedMore):
                    if not rest:
                        raise RuntimeError("ERROR: Ran out of data mid-decompression.")
                    st = st.feed(rest.pop(0))
                elif isinstance(st, Chunk):
                    got.append(st.chunk)
                    st = st.next()
                elif isinstance(st, Done):
                    if rest:
                        raise RuntimeError("ERROR: Finished decompression with data left.")
                    break
                elif isinstance(st, DecompError):
                    raise RuntimeError("ERROR: " + st.error.show())
            if b"".join(got) != gold:
                raise RuntimeError("output differs from .gold")
        return run

    def incremental_zlib(chunks: List[bytes], gold: bytes):
        # Benchmark.hs:72-87 decompressIncrementalC
        def run():
            d = _czlib.decompressobj()
            got = [d.decompress(c) for c in chunks]
            got.append(d.flush())
            if d.unused_data:
                raise RuntimeError("ERROR: Finished decompression with data left.")
            if b"".join(got) != gold:
                raise RuntimeError("output differs from .gold")
        return run

    for tc in cases:
        z, gold = get_files(directory, tc)
        chunks = lazy_chunks(z)

        def normal_pzgpu(chunks=chunks, gold=gold):
            r = pz.decompress(chunks, ctx=ctx, size_hint=len(gold))
            if not isinstance(r, pz.Right) or r.value != gold:
                raise RuntimeError(f"pzgpu result differs from .gold: {r!r:.80}")

        def normal_zlib(z=z, gold=gold):
            if _czlib.decompress(z) != gold:
                raise RuntimeError("output differs from .gold")

        out.append((f"decompression/{tc}/normal/pzgpu", normal_pzgpu, len(gold)))
        out.append((f"decompression/{tc}/normal/zlib", normal_zlib, len(gold)))
        out.append((f"decompression/{tc}/incremental/pzgpu", incremental_pzgpu(chunks, gold), len(gold)))
        out.append((f"decompression/{tc}/incremental/zlib", incremental_zlib(chunks, gold), len(gold)))
    for tc in cases:
        z, gold = get_files(directory, tc)
        for n in batch_sizes:
            if n * len(gold) > (2 << 30):
                continue  # keep a sample within a couple of GiB of host memory

            def many(z=z, gold=gold, n=n):
                rs = pz.decompress_many([z] * n, ctx=ctx, size_hint=[len(gold)] * n)
                if not all(isinstance(r, pz.Right) and r.value == gold for r in rs):
                    raise RuntimeError("a batch member differs from .gold")
            out.append((f"batch/{tc}/n={n}/pzgpu", many, n * len(gold)))
    return out


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="python -m pure_zlib_amd.benchmark", description=__doc__.split("\n\n")[0])
    ap.add_argument("--dir", default=DEFAULT_DIR, help="directory of <case>.z / <case>.gold pairs")
    ap.add_argument("--cases", default="", help="comma-separated case names (default: every pair in --dir)")
    ap.add_argument("--time-limit", type=float, default=1.0, help="seconds of sampling per benchmark (criterion -L)")
    ap.add_argument("--batch", default="1,64,4096", help="decompressMany batch sizes to sweep ('' = none)")
    ap.add_argument("--match", default="", help="only benchmarks whose name contains this (criterion's pattern)")
    ap.add_argument("--list", action="store_true", help="print the benchmark names and exit (criterion --list)")
    args = ap.parse_args(argv)
    cases = [c for c in args.cases.split(",") if c] or find_cases(args.dir)
    batch = [int(b) for b in args.batch.split(",") if b]
    if args.list:
        for tc in cases:
            for grp in ("normal", "incremental"):
                for impl in ("pzgpu", "zlib"):
                    print(f"decompression/{tc}/{grp}/{impl}")
        for tc in cases:
            for n in batch:
                print(f"batch/{tc}/n={n}/pzgpu")
        return 0
    for name, thunk, nbytes in build_benchmarks(args.dir, cases, batch):
        if args.match and args.match not in name:
            continue
        r = measure(thunk, args.time_limit)
        print(f"benchmarking {name}\n  time {fmt_time(r['mean'])}  (min {fmt_time(r['min'])}, std dev {fmt_time(r['stddev'])}, "
              f"{int(r['samples'])} samples)   {nbytes / r['mean'] / 2**20:10.1f} MiB/s decoded, full output checked")
    return 0


if __name__ == "__main__":
    sys.exit(main())
