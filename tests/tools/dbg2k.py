import sys, os, zlib, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch; torch.cuda.init()
import pure_zlib_amd as P
import bench
class A: pass
import argparse
ns = argparse.Namespace(pool=4096, blob_bytes=2048, workload="l6_32k", level=6, gzip=False)
texts, zs = bench.build_pool(ns)
ctx = P.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(1)
pick = rng.integers(0, len(zs), size=n)
streams = [zs[i] for i in pick]
res = P.decompress_many(streams, ctx=ctx, size_hint=[2048] * n)
bad = 0
for k, r in enumerate(res):
    exp = texts[pick[k]]
    got = r.value if hasattr(r, "value") else r
    if not isinstance(r, P.Right) or r.value != exp:
        bad += 1
        if bad <= 5:
            if isinstance(r, P.Right):
                g = r.value
                m = next((i for i in range(min(len(g), len(exp))) if g[i] != exp[i]), None)
                print("stream", k, "pool", pick[k], "len", len(g), "exp", len(exp), "first mismatch", m, "zlen", len(streams[k]))
            else:
                print("stream", k, "pool", pick[k], "Left", r.value.show(), "zlen", len(streams[k]))
print("bad", bad, "of", n)
