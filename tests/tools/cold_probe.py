"""Lab tool: kernel ms of the first four launches of the headline batch in a fresh process -- as they come (plain), behind a launch of 64
streams (small_first: the code object, the clocks) or behind one small stream per resident stream-wave (full_small_first: every wave's
scratch touched and a profile learned).  Round 6, one box: 7.85 / 6.43 / 6.21 / 6.19, 7.01 / 6.43 / 6.26 / 6.24, 6.94 / 6.35 / 6.17 / 6.19:
~0.8 ms of the cold launch is the process' first kernel, the rest is gone by the third launch.  python tests/tools/cold_probe.py <mode>"""
import sys, os, zlib, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torch
import corpus
import pure_zlib_amd as P
from devbatch import DeviceBatch
mode = sys.argv[1]
tv = [corpus.zipf_text(32768, s) for s in range(512)]
zv = [zlib.compress(t, 6) for t in tv]
pick = np.random.default_rng(1).integers(0, len(zv), size=65536)
vb = DeviceBatch(tv, zv, pick)
small = DeviceBatch(tv[:64], zv[:64], np.arange(64))
ctx = P.Context(0)
if mode == "small_first":
    small.run(ctx, 11); print("small", round(ctx.last_kernel_ms(), 3))
if mode == "full_small_first":   # as many streams as the chip holds waves, tiny: touches every wave's scratch
    pk = np.arange(6656) % 64
    s2 = DeviceBatch(tv[:64], zv[:64], pk)
    s2.run(ctx, 11); print("6656 streams", round(ctx.last_kernel_ms(), 3))
ms = []
for i in range(int(os.environ.get("COLD_N", "4"))):
    vb.run(ctx, 11); ms.append(round(ctx.last_kernel_ms(), 3))
print(mode, ms)
