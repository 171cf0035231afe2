import sys, zlib, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch; torch.cuda.init()
import corpus
import pure_zlib_amd as P
from pure_zlib_amd import benchmark as HB
ctx = P.Context(0)
plain = [corpus.zipf_text(256 * 1024, 7000 + k) for k in range(32)]
zs = [zlib.compress(t, 6) for t in plain]
for n in (64, 1024, 4096):
    t0 = time.time(); r = HB.incremental_throughput(ctx, zs, plain, n_decoders=n); print(n, r, "wall %.1fs" % (time.time() - t0))
small = [corpus.zipf_text(20000, k) for k in range(32)]
r = HB.incremental_throughput(ctx, [zlib.compress(t, 6) for t in small], small, n_decoders=4096, piece=4096, room=65536); print("small", r)
