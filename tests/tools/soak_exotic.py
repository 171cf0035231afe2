"""Lab tool (GPU box): thousands of writer-made streams (tests/deflate_writer.py) and zlib-made strip cases, valid and corrupted,
through the GPU path against the oracle -- more seeds than the test-suite takes.  Usage: python tests/tools/soak_exotic.py [first] [count] [ring]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.cuda.init()
import corpus  # noqa: E402
import deflate_writer as W  # noqa: E402
import pure_zlib_amd as P  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_parity import _check_against_oracle, run_batch  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ring = int(sys.argv[3]) if len(sys.argv) > 3 else 11
ctx = P.Context(0)
ctx.set_ring_bits(ring)
done = 0
for lo in range(first, first + count, 100):
    streams, caps, datas = [], [], []
    for seed in range(lo, min(lo + 100, first + count)):
        d, z = (W.exotic_stream(seed)[:2] if seed % 3 else corpus.strip_case(seed))
        streams.append(z); caps.append(len(d)); datas.append(d)
        for c in range(4):
            streams.append(corpus.corrupt(z, seed * 16 + c)); caps.append([len(d) + 64, len(d) // 2, len(d)][c % 3]); datas.append(None)
    res, outs, _, _ = run_batch(ctx, streams, caps)
    _check_against_oracle(O, streams, caps, res, outs, datas)
    done += len(streams)
    print(f"seeds {lo}..: {done} streams ok", flush=True)
ctx.close()
print("soak ok", done, "ring", ring)
