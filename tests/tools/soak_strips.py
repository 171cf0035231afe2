"""Diagnostic soak (not part of the suites): several-span streams of every kind (tests/corpus.py strip_case), binary-looking records (corpus.binary_records) and corrupted variants
of each through the GPU path on rings 11 / 13 / 15, against the oracle.  python tests/tools/soak_strips.py [n] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import pure_zlib_amd as P
from oracle import oracle as O
from test_gpu_parity import run_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = P.Context(0)
streams, caps = [], []
for seed in range(seed0, seed0 + n):
    d, z = corpus.strip_case(seed)
    streams.append(z); caps.append(len(d))
    for c in range(5):
        streams.append(corpus.corrupt(z, seed * 16 + c)); caps.append([len(d) + 64, len(d) // 2, len(d)][c % 3])
nbin = 0
for seed in range(seed0, seed0 + n // 2):  # (round 6) binary-looking records: long codes in constant use, resolved inside the spans
    import zlib
    d = corpus.binary_records([8, 24, 30, 64, 100, 33][seed % 6] * 1024, seed)
    z = zlib.compress(d, 1 + seed % 9)
    streams.append(z); caps.append(len(d)); nbin += 1
    for c in range(5):
        streams.append(corpus.corrupt(z, seed * 16 + c)); caps.append([len(d) + 64, len(d) // 2, len(d)][c % 3])
ref = [O.decompress(z, cap) for z, cap in zip(streams, caps)]
for rb in (11, 13, 15):
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
    print(f"ring {rb}: {len(streams)} streams ({n + nbin} valid, {nbin} of them binary records), mismatches {len(bad)} {bad[:5]}")
