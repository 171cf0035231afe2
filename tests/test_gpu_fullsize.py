"""GPU parity at BASELINE.json's FULL sizes, one launch each, through the C ABI with the arenas
resident in HBM exactly as bench.py's timed region hands them over:

  config 3   65,536 x 4 KiB fixed-Huffman (Z_FIXED, level 1) blobs  -- rings 11 and 15
  config 5   131,072 mixed 1-64 KiB level-6 blobs = its per-GPU share (1 M streams over 8 GPUs)
  N > 1      bench.py ITSELF under torch.distributed.run with two ranks (both on device 0, gloo rendezvous:
             RCCL refuses two ranks on one GPU): the sharding, the barrier, MAX of the time, MIN of bit_exact

Every stream: status, length, in_used, Adler-32 against zlib.adler32, every decoded byte against the text it was
compressed from, and 256 sampled streams against the oracle (the restatement of the reference)."""
import json
import os
import socket
import subprocess
import sys
import zlib

import numpy as np
import pytest

import corpus
from conftest import ROOT
from devbatch import DeviceBatch

pytestmark = pytest.mark.gpu


def _fixed_pool(npool):
    texts, zs = [], []
    for seed in range(npool):
        t = corpus.zipf_text(4096, seed)
        co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        texts.append(t)
        zs.append(co.compress(t) + co.flush())
    return texts, zs


def test_config3_65536_fixed_huffman_4k_blobs(gpu_ctx, oracle):
    """BASELINE config 3 at full size.  A persistent wave decodes ~10 streams back to back here, so the
    fixed tables built once per wave (WaveLds::fixed_ready) are reused across streams."""
    texts, zs = _fixed_pool(1024)
    assert all((z[2] >> 1) & 3 == 1 for z in zs)  # BTYPE 01: fixed Huffman
    pick = np.random.default_rng(0xC3).integers(0, len(zs), size=65536)
    b = DeviceBatch(texts, zs, pick)
    for ring in (11, 15):
        res = b.run(gpu_ctx, ring)
        b.check_all(*res)
        b.check_sample_vs_oracle(oracle, 256)
    gpu_ctx.set_ring_bits(11)


def test_262144_small_level6_blobs_2k(gpu_ctx, oracle):
    """4,096 distinct 2 KiB level-6 blobs, 64 replicas each, in one launch (the profile sweeps' 1 M x 2 KiB batch at a
    quarter of its size).  Small dynamic-Huffman streams end their windows at a stopper with the token queue nearly
    full far more often than long ones: round 2 found the queue one token over its capacity here (blob 738)."""
    texts = [corpus.zipf_text(2048, seed) for seed in range(4096)]
    zs = [zlib.compress(t, 6) for t in texts]
    pick = np.random.default_rng(1).integers(0, len(zs), size=262144)
    b = DeviceBatch(texts, zs, pick)
    for ring in (11, 12):
        res = b.run(gpu_ctx, ring)
        b.check_all(*res)
    b.check_sample_vs_oracle(oracle, 128)
    gpu_ctx.set_ring_bits(11)


def test_config5_per_gpu_share_131072_mixed_blobs(gpu_ctx, oracle):
    """BASELINE config 5's per-GPU share: 1 M mixed 1-64 KiB level-6 blobs over 8 GPUs = 131,072 per GPU
    (4 GiB decoded per launch), laid out longest first as the sharder hands a shard over."""
    from pure_zlib_amd.shard import plan_shards
    texts, zs = [], []
    for seed in range(1024):
        size = 1024 * (1 + (seed * 2654435761 >> 7) % 64)
        t = corpus.zipf_text(size, seed)
        texts.append(t)
        zs.append(zlib.compress(t, 6))
    assert {len(t) for t in texts} == {1024 * k for k in range(1, 65)}
    perm = np.random.default_rng(0xC5).integers(0, len(zs), size=1 << 20)  # the whole node's batch
    dec_len = np.array([len(t) for t in texts], dtype=np.int64)
    shards = plan_shards(dec_len[perm], 8)
    mine = shards[3]
    assert abs(len(mine) - 131072) < 2048
    b = DeviceBatch(texts, zs, perm[mine])
    res = b.run(gpu_ctx, 11)
    b.check_all(*res)
    b.check_sample_vs_oracle(oracle, 256)


def _bench_two_ranks(extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
               PZG_BENCH_DEVICE="0", PZG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pool", "256"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # ONE line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["bit_exact"] is True and r["scaling"] == "weak"
    return r


def test_bench_py_two_ranks_on_one_device():
    """The real N>1 path of bench.py: torch.distributed.run launches two ranks before anything touches the GPU; each
    decodes its own shard through libpzg.so, the time is MAX-reduced, bit_exact MIN-reduced, rank 0 prints the line."""
    r = _bench_two_ranks(["--streams", "4096"])
    assert r["config"]["streams_per_gpu"] == 4096 and r["config"]["parallelism"] == "shard2"
    # whole-job value: both ranks' bytes over the max-over-ranks time
    assert abs(r["value"] - 2 * 4096 * 32768 / (r["ms_per_step"] * 1e-3) / 2**30) / r["value"] < 0.02
    assert "cpu_baseline" not in r and r["roofline"]["frac"] > 0


def test_bench_py_config5_command_two_ranks():
    """BASELINE config 5's command -- `bench.py --gpus 8 --workload mixed --streams 131072` (131,072 mixed 1-64 KiB level-6
    streams per GPU, 1 M on the node) -- with two ranks folded onto device 0 and a smaller per-rank share: the LPT shard
    plan over mixed sizes, every stream compared with its plaintext on each rank."""
    r = _bench_two_ranks(["--workload", "mixed", "--streams", "16384"])
    assert 14000 < r["config"]["streams_per_gpu"] < 19000 and r["config"]["parallelism"] == "shard2"  # (balanced by bytes, not by count)
    assert "config 5" in r["config"]["workload"] and r["value"] > 0


def test_adler32_over_one_9_gib_device_buffer(gpu_ctx):
    """BASELINE config 2 beyond 4 GiB under -m gpu: one 9 GiB device buffer, SURVEY.md 8d's splitmix64 bytes,
    against chunked zlib.adler32 over the CPU generator's bytes (Adler32.hs:37-57: the combine step multiplies a block's
    byte sum by the length that follows it, a 64-bit number here, modulo 65521).  Also an unaligned start and an odd length."""
    import torch
    dev = torch.device("cuda", 0)
    nwords = (9 << 30) // 8
    buf = torch.empty(nwords, dtype=torch.int64, device=dev)
    chunk = 1 << 25  # words: 256 MiB
    exp, exp_odd = 1, 1
    skew, tail_cut = 5, 3  # the second checksum runs over bytes [5, 9 GiB - 3)
    for lo in range(0, nwords, chunk):
        cnt = min(chunk, nwords - lo)
        buf[lo:lo + cnt] = corpus.splitmix64_torch(lo, cnt, dev)
        host = corpus.splitmix64_numpy(lo, cnt).view(np.uint8)
        if lo in (0, (nwords // chunk // 2) * chunk):  # device fill == CPU generator (whole chunks, two of them)
            assert np.array_equal(buf[lo:lo + cnt].cpu().numpy().view(np.uint8), host)
        exp = zlib.adler32(host, exp)
        a = skew if lo == 0 else 0
        b = len(host) - (tail_cut if lo + cnt == nwords else 0)
        exp_odd = zlib.adler32(host[a:b], exp_odd)
    res = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.adler32_device(buf.data_ptr(), nwords * 8, res.data_ptr(), sync=True)
    gpu_ctx.adler32_device(buf.data_ptr() + skew, nwords * 8 - skew - tail_cut, res.data_ptr() + 4, sync=True)
    got = res.cpu().numpy().view(np.uint32)
    assert int(got[0]) == exp and int(got[1]) == exp_odd, (hex(int(got[0])), hex(exp), hex(int(got[1])), hex(exp_odd))
    del buf
