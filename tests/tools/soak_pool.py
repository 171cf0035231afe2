"""Diagnostic soak (not part of the suites): writer-made codes around the second-level pool's size (tests/deflate_writer.py pool_stream)
and corrupted variants through the GPU path on rings 11 / 12 / 15, against the oracle.  python tests/tools/soak_pool.py [n] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import deflate_writer as W
import pure_zlib_amd as P
from oracle import oracle as O
from test_gpu_parity import run_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = P.Context(0)
streams, caps = [], []
for seed in range(seed0, seed0 + n):
    d, z = W.pool_stream(seed)
    streams.append(z); caps.append(len(d))
    for c in range(3):
        streams.append(corpus.corrupt(z, seed * 8 + c)); caps.append([len(d) + 64, len(d) // 2, len(d)][c])
ref = [O.decompress(z, cap) for z, cap in zip(streams, caps)]
for rb in (11, 12, 15):
    ctx.set_ring_bits(rb)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, caps)
    bad = []
    for k, (r, o) in enumerate(ref):
        if status[k] != r.status:
            bad.append(k)
        elif r.status == 0 and (outs[k] != o or int(adler[k]) != r.adler or int(in_used[k]) != r.in_used):
            bad.append(k)
        elif r.status == 14 and int(out_len[k]) != r.out_len:
            bad.append(k)
        elif r.status in (3, 4, 6, 10, 11, 12, 13) and [int(detail[k][0]), int(detail[k][1])] != [r.detail0, r.detail1]:
            bad.append(k)
    print(f"ring {rb}: {len(streams)} streams ({n} valid), mismatches {len(bad)} {bad[:5]}")
