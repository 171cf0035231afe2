import sys, os, zlib, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
import bench, corpus
import pure_zlib_amd as P
from devbatch import DeviceBatch
ctx = P.Context(0)
for kind in range(4):
    for lvl_idx in range(3):
        seeds = [s for s in range(768) if s % 4 == kind and (s // 16) % 3 == lvl_idx][:48]
        hv = [bench.hetero_blob(s) for s in seeds]
        tv, zv = [h[0] for h in hv], [h[1] for h in hv]
        pick = np.random.default_rng(1).integers(0, len(zv), size=8192)
        vb = DeviceBatch(tv, zv, pick)
        vb.check_all(*vb.run(ctx, 11))
        ms = []
        for _ in range(3):
            vb.run(ctx, 11); ms.append(ctx.last_kernel_ms())
        dec = int(vb.out_cap.sum())
        print("kind", kind, "level", [6,1,9][lvl_idx], "GiB/s", round(dec / (np.mean(ms) * 1e-3) / 2**30, 1), "ms", round(float(np.mean(ms)), 2), "ratio", round(dec / int(vb.in_len.sum()), 2), flush=True)
