"""Diagnostic soak (not part of the suites): many resumable decoders fed random piece sizes through DecoderPool.feed (several
decoders per launch, mixed progress), every decoder's whole event trace (NeedMore / Chunk(len) / Done / DecompError) and its
bytes against the oracle's restatement of Monad.hs:163-197.  python tests/tools/soak_incremental.py [streams] [seed0]"""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.cuda.init()
import corpus
import pure_zlib_amd as P
from pure_zlib_amd.incremental import Chunk, DecoderPool, DecompError, Done, NeedMore
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
ctx = P.Context(0)
bad = 0
B = 96  # decoders per pool
for base in range(0, n, B):
    seeds = list(range(seed0 + base, seed0 + min(n, base + B)))
    rng = random.Random(seeds[0])
    room = rng.choice([4096, 20000, 70000, 300000])
    pool = DecoderPool(len(seeds), ctx, room=room)
    streams, pieces = [], []
    for s in seeds:
        size = [0, 1, 300, 5000, 40000, 90000, 250000][s % 7]
        d = corpus.mixed_data(size, s) if s % 3 else corpus.zipf_text(size, s)
        z = corpus.compress_variant(d, s) if s % 2 else zlib.compress(d, 1 + s % 9)
        if s % 9 == 0:
            z = corpus.corrupt(z, s)
        r = random.Random(s)
        ps, i = [], 0
        while i < len(z):
            k = r.choice([1, 3, 50, 700, 7000, 40000])
            ps.append(z[i:i + k]); i += k
        if s % 13 == 0 and ps:
            ps.insert(len(ps) // 2, b"")
        streams.append(z); pieces.append(ps)
    events = [[("NeedMore",)] for _ in seeds]
    data = [bytearray() for _ in seeds]
    pos = [0] * len(seeds)
    alive = [True] * len(seeds)
    while any(alive[k] and pos[k] < len(pieces[k]) for k in range(len(seeds))):
        ks = [k for k in range(len(seeds)) if alive[k] and pos[k] < len(pieces[k]) and rng.random() < 0.8]
        if not ks:
            continue
        sts = pool.feed(ks, [pieces[k][pos[k]] for k in ks])
        for k, st in zip(ks, sts):
            pos[k] += 1
            while isinstance(st, Chunk):
                events[k].append(("Chunk", len(st.chunk))); data[k] += st.chunk
                st = st.next()
            if isinstance(st, NeedMore):
                events[k].append(("NeedMore",))
            elif isinstance(st, Done):
                events[k].append(("Done",)); alive[k] = False
            else:
                events[k].append(("DecompError", st.error.show())); alive[k] = False
    for k, s in enumerate(seeds):
        eo, ro, oo = O.trace(pieces[k])
        want = [e if e[0] != "DecompError" else ("DecompError", ro.message.decode()) for e in eo]
        got = events[k]
        if got != want or (ro.status == 0 and bytes(data[k]) != oo):
            bad += 1
            if bad <= 5:
                print("MISMATCH seed", s, "room", room, "status", ro.status, got[-3:], want[-3:])
    pool.close()
print(f"incremental soak: {n} decoders, mismatches {bad}")
