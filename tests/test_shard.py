"""CPU: the multi-GPU form of decompressMany is a partition of the stream list (no collective on the
data path).  The planner is checked directly and through a world_size-2 gloo run."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT
from pure_zlib_amd.shard import plan_shards, shard_imbalance


def test_plan_is_a_partition_and_balanced():
    rng = np.random.default_rng(1)
    w = 1024 * rng.integers(1, 65, size=20000)
    for world in (1, 2, 3, 8):
        sh = plan_shards(w, world)
        allidx = np.sort(np.concatenate(sh))
        assert (allidx == np.arange(len(w))).all()
        assert shard_imbalance(w, sh) < 1.01
    u = np.full(65536, 32768)
    sh = plan_shards(u, 8)
    assert all(len(s) == 8192 for s in sh) and (sh[3] == np.arange(3 * 8192, 4 * 8192)).all()
    assert [len(s) for s in plan_shards([], 4)] == [0, 0, 0, 0]


WORKER = r'''
import os, sys, zlib, hashlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import corpus
from pure_zlib_amd.shard import plan_shards
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# the same global batch on every rank (seeded); each rank decodes only its shard.  On this CPU-only box
# the decode stand-in is the oracle: what is under test is the sharding + timing plumbing of bench.py.
sizes = [1024 * (1 + (i * 7) % 16) for i in range(48)]
streams = [zlib.compress(corpus.zipf_text(n, i), 6) for i, n in enumerate(sizes)]
mine = plan_shards(sizes, world)[rank]
digest = np.zeros(48, dtype=np.int64)
for i in mine:
    r, out = O.decompress(streams[i], sizes[i])
    assert r.status == 0
    digest[i] = r.adler
t = torch.from_numpy(digest)
dist.barrier()
dist.all_reduce(t, op=dist.ReduceOp.SUM)          # each stream decoded by exactly one rank
elapsed = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)    # bench.py: max over ranks
exp = np.array([zlib.adler32(corpus.zipf_text(n, i)) for i, n in enumerate(sizes)], dtype=np.int64)
assert (t.numpy() == exp).all(), "a stream was skipped or decoded twice"
assert abs(elapsed.item() - 0.1 * world) < 1e-9
if rank == 0:
    print("SHARD_OK", world, len(mine))
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "SHARD_OK 2 24" in out.stdout


def test_benchmark_harness_listing_and_sampling(capsys):
    """The harness mirror (Benchmark.hs:12-46) without a GPU: case discovery, group names, the sampler."""
    from pure_zlib_amd import benchmark
    cases = benchmark.find_cases(benchmark.DEFAULT_DIR)
    assert cases == ["randtest1", "randtest2", "randtest3", "rfctest1", "rfctest2", "rfctest3",
                     "zerotest1", "zerotest2", "zerotest3"]  # Benchmark.hs:12-24 minus tor-list (not in the reference tree)
    assert benchmark.main(["--list", "--cases", "rfctest1", "--batch", "1,64"]) == 0
    out = capsys.readouterr().out.split()
    assert out == ["decompression/rfctest1/normal/pzgpu", "decompression/rfctest1/normal/zlib",
                   "decompression/rfctest1/incremental/pzgpu", "decompression/rfctest1/incremental/zlib",
                   "batch/rfctest1/n=1/pzgpu", "batch/rfctest1/n=64/pzgpu"]
    z, gold = benchmark.get_files(benchmark.DEFAULT_DIR, "rfctest1")
    assert len(benchmark.lazy_chunks(z)) == (len(z) + 32767) // 32768
    calls = []
    r = benchmark.measure(lambda: calls.append(1), 0.01)
    assert r["samples"] >= 3 and len(calls) == int(r["samples"]) + 1 and r["min"] <= r["mean"]
