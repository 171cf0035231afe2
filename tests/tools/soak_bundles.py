"""Lab tool (GPU box): thousands of streams of the FIXED code -- zlib-made (Z_FIXED, levels 1 / 6 / 9, 1..17 blocks) and writer-made
(tests/deflate_writer.py: fixed blocks with chosen tokens), valid and corrupted, capacities exact / too small / generous -- through the
bundles (PZG_OPT_BUNDLES 2) against the oracle.  Usage: python tests/tools/soak_bundles.py [first] [count]"""
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.cuda.init()
import corpus  # noqa: E402
import deflate_writer as W  # noqa: E402
import pure_zlib_amd as P  # noqa: E402
from oracle import oracle as O  # noqa: E402
import numpy as np  # noqa: E402
from devbatch import DeviceBatch  # noqa: E402
from test_gpu_bundles import _check_against_oracle  # noqa: E402
from test_model_bundles import fixed  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000


def writer_fixed(seed):
    """a stream of 1-6 fixed blocks made by the writer: any length / distance the format allows, distance-1 runs, 258-byte matches"""
    rng = random.Random(0xF1 + seed)
    out = bytearray()
    w = W.BitWriter()
    nb = rng.randint(1, 6)
    for i in range(nb):
        b = W.Block("fixed")
        b.tokens = W.gen_tokens(rng, out, rng.choice([1, 40, 700, 3000, 9000]), dict(alphabet=W._alphabet(rng, rng.choice(["text", "wide", "four", "any"])),
                                                                                    lens=rng.choice([list(range(3, 259)), [3, 4, 5, 258], [3, 3, 4, 6, 9]]),
                                                                                    dists=rng.choice(["any", "near", "far", "one", "max"]), p_match=rng.choice([0.02, 0.3, 0.6])))
        b.opts = {}
        W.write_block(w, b, i == nb - 1, rng, b.opts)
    data = bytes(out)
    return data, bytes([0x78, 0x01]) + w.bytes() + zlib.adler32(data).to_bytes(4, "big")


ctx = P.Context(0)
ctx.set_bundles(2)
done = 0
for lo in range(first, first + count, 200):
    streams, caps, datas = [], [], []
    for seed in range(lo, min(lo + 200, first + count)):
        rng = random.Random(seed)
        kind = seed % 6
        if kind == 0:
            d, z = writer_fixed(seed)
        else:
            n = rng.choice([0, 1, 17, 300, 1000, 4096, 4097, 20000, 70000]) if kind == 1 else rng.randrange(1, 8200)
            d = [corpus.zipf_text, corpus.html_slice, corpus.skewed_bytes, corpus.mixed_data, corpus.zipf_text][kind - 1](n, seed)
            if kind == 3:
                d = bytes(b % 144 for b in d)
            z = fixed(d, level=rng.choice([1, 6, 9]), blocks=rng.choice([1, 1, 2, 5, 17]))
        streams.append(z); caps.append(len(d)); datas.append(d)
        for c in range(3):
            streams.append(corpus.corrupt(z, seed * 16 + c)); caps.append([len(d) + 64, len(d) // 2, len(d)][c % 3]); datas.append(None)
    # (a device-pointer launch: the bundles take those; a "text" of the capacity's length stands for every stream, the oracle says what is right)
    b = DeviceBatch([bytes(c) for c in caps], streams, np.arange(len(streams)))
    res = b.run(ctx, 11)
    _check_against_oracle(b, res, O)
    for k, d in enumerate(datas):
        if d is not None:
            assert res[0][k] == 0 and bytes(b.d_out[int(b.out_off[k]):int(b.out_off[k]) + len(d)].cpu().numpy()) == d, k
    done += len(streams)
    print(f"seeds {lo}..: {done} streams ok", flush=True)
ctx.close()
print("bundle soak ok", done)
