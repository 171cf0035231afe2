"""Diagnostic soak (not part of the suites): many seeded valid and corrupted streams of every corpus through the
GPU path, valid ones against system zlib, corrupted ones against the oracle.  python tests/tools/soak.py [n] [seed0]"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import pure_zlib_amd as P
import pure_zlib_amd.zlib as Z
from oracle import oracle as O
from test_gpu_parity import run_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ctx = P.Context(0)
gens = [corpus.mixed_data, corpus.html_slice, corpus.skewed_bytes, corpus.zipf_text, corpus.random_bytes]
for rb in ((11, 12, 15) if not os.environ.get('SOAK_GZIP_ONLY') else ()):
    ctx.set_ring_bits(rb)
    streams, datas = [], []
    for k in range(n):
        seed = seed0 + k
        size = [0, 1, 7, 300, 3000, 9000, 33000, 66000, 140000][seed % 9] if seed % 4 == 0 else (seed * 2654435761 >> 8) % 30000
        if rb == 15:
            size = min(size, 40000)
        d = gens[seed % len(gens)](min(size, 120000), seed)
        streams.append(corpus.compress_variant(d, seed) if seed % 2 else zlib.compress(d, seed % 10))
        datas.append(d)
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, streams, [len(d) for d in datas])
    bad = [k for k in range(n) if status[k] != 0 or outs[k] != datas[k] or int(adler[k]) != zlib.adler32(datas[k]) or int(in_used[k]) != len(streams[k])]
    print(f"ring {rb}: valid {n} streams, mismatches {len(bad)} {bad[:5]}")
    m = n // 2
    cor = [corpus.corrupt(streams[k], seed0 + 7 * k) for k in range(m)]
    caps = [len(datas[k]) + (64 if k % 3 else 0) for k in range(m)]
    (out_len, status, detail, in_used, adler), outs, _, _ = run_batch(ctx, cor, caps)
    nbad = 0
    for k in range(m):
        r, o = O.decompress(cor[k], caps[k])
        ok = int(status[k]) == r.status
        if ok and r.status == 0:
            ok = outs[k] == o and int(adler[k]) == r.adler
        elif ok and r.status == 14:
            ok = int(out_len[k]) == r.out_len
        elif ok:
            ok = Z.error_from_status(cor[k], int(status[k]), detail[k]).show() == r.message.decode()
        if not ok:
            nbad += 1
            if nbad < 5:
                print("  corrupt mismatch", k, int(status[k]), r.status, r.message)
    print(f"ring {rb}: corrupted {m} streams, mismatches {nbad}")

# gzip members (extension): valid ones against zlib (wbits 31), corrupted ones against the oracle's gzip restatement
ctx.set_ring_bits(11)
m = n // 2
gz_d = [gens[(seed0 + k) % len(gens)](((seed0 + k) * 40503 >> 4) % 50000, seed0 + k) for k in range(m)]
gz_z = [corpus.gzip_member(d, seed0 + k) for k, d in enumerate(gz_d)]
(out_len, status, detail, in_used, crc), outs, _, _ = run_batch(ctx, gz_z, [len(d) for d in gz_d], gzip=True)
bad = [k for k in range(m) if status[k] != 0 or outs[k] != gz_d[k] or int(crc[k]) != zlib.crc32(gz_d[k]) or int(in_used[k]) != len(gz_z[k])]
print(f"gzip: valid {m} members, mismatches {len(bad)} {bad[:5]}")
cor = [corpus.corrupt(gz_z[k], seed0 + 11 * k) for k in range(m)]
caps = [len(d) + 4096 for d in gz_d]
(out_len, status, detail, in_used, crc), outs, _, _ = run_batch(ctx, cor, caps, gzip=True)
nbad = 0
for k in range(m):
    r, o = O.gzip_decompress(cor[k], caps[k])
    if int(status[k]) == 14 and r.status in (10, 19):
        continue
    ok = int(status[k]) == r.status and (r.status != 0 or (outs[k] == o and int(crc[k]) == r.adler))
    if not ok:
        nbad += 1
        if nbad < 5:
            print("  gzip corrupt mismatch", k, int(status[k]), r.status, r.message)
print(f"gzip: corrupted {m} members, mismatches {nbad}")
