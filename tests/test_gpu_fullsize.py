"""GPU parity at BASELINE.json's FULL sizes, one launch each, through the C ABI with the arenas
resident in HBM exactly as bench.py's timed region hands them over:

  config 3   65,536 x 4 KiB fixed-Huffman (Z_FIXED, level 1) blobs  -- rings 11 and 15
  config 4   65,536 x 32 KiB dynamic-Huffman (level 6) blobs         -- rings 11 and 15
  config 5   131,072 mixed 1-64 KiB level-6 blobs = its per-GPU share (1 M streams over 8 GPUs)
  config 2   Adler-32 over one 16 GiB buffer
  > 4 GiB    one stream that decodes to more than 4 GiB and one whose COMPRESSED form is more than 4 GiB (include/pzg.h:
             "one compressed stream up to 16 GiB ... the decoded size is not limited")
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
    try:
        # (round 6: a launch of this size takes its streams of the fixed code 64 to a wave, one lane per stream -- bundle_core.h; with
        # PZG_OPT_BUNDLES 0 every stream goes to the one-stream-per-wave kernel as before)
        for bundles in (1, 0):
            gpu_ctx.set_bundles(bundles)
            for ring in (11, 15):
                res = b.run(gpu_ctx, ring)
                b.check_all(*res)
                b.check_sample_vs_oracle(oracle, 256)
    finally:
        gpu_ctx.set_bundles(1)
        gpu_ctx.set_ring_bits(11)


def test_config4_65536_level6_32k_blobs(gpu_ctx, oracle):
    """BASELINE config 4 at full size (the batch bench.py times), under -m gpu: 65,536 x 32 KiB level-6 blobs, every
    stream's status / length / in_used / Adler-32, every decoded byte, 256 sampled streams against the oracle -- on the
    default hybrid ring and on north_star's literal 32 KiB LDS ring."""
    texts = [corpus.zipf_text(32768, seed) for seed in range(2048)]
    zs = [zlib.compress(t, 6) for t in texts]
    pick = np.random.default_rng(0xC4).integers(0, len(zs), size=65536)
    b = DeviceBatch(texts, zs, pick)
    for ring in (11, 15):
        res = b.run(gpu_ctx, ring)
        b.check_all(*res)
        b.check_sample_vs_oracle(oracle, 256)
    gpu_ctx.set_ring_bits(11)


def test_config4_with_the_scratch_capped_at_64_mib(gpu_ctx, oracle):
    """PZG_OPT_SCRATCH_BYTES bounds the device memory the library takes for its kernels' scratch (419 MiB per
    arena for a launch that fills the chip, up to six arenas per device).  With 64 MiB for the whole device (10.7 MiB per arena:
    169 of the 6,656 stream-waves own a slice, the others decode by windows) BASELINE config 4 at full size is still bit-exact --
    every stream, every byte, 256 sampled streams against the oracle; so is a launch with no scratch at all (a 1-byte cap), the
    host-pointer path under the cap, and the next launch after the cap is lifted."""
    import pure_zlib_amd as P
    texts = [corpus.zipf_text(32768, seed) for seed in range(1024)]
    zs = [zlib.compress(t, 6) for t in texts]
    pick = np.random.default_rng(0xC8).integers(0, len(zs), size=65536)
    b = DeviceBatch(texts, zs, pick)
    try:
        for cap in (64 << 20, 1, 0):
            gpu_ctx.set_scratch_bytes(cap)
            res = b.run(gpu_ctx, 11)
            b.check_all(*res)
            if cap:
                b.check_sample_vs_oracle(oracle, 128 if cap > 1 else 16)
                got = P.decompress_many(zs[:256], ctx=gpu_ctx)  # the host-pointer path (its own arena) under the cap
                assert all(g == P.Right(t) for g, t in zip(got, texts[:256]))
    finally:
        gpu_ctx.set_scratch_bytes(0)
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
    # every rank's own time and share beside the job's (the job's time is the slowest rank's)
    pr = r["per_rank"]
    assert len(pr["kernel_ms"]) == 2 and len(pr["wall_ms_per_step"]) == 2 and pr["streams"] == [4096, 4096]
    assert pr["kernel_ms_min"] <= pr["kernel_ms_max"] and abs(max(pr["wall_ms_per_step"]) - r["ms_per_step"]) < 1e-3
    assert 1.0 <= pr["shard_imbalance"] < 1.01


def test_bench_py_config5_command_two_ranks():
    """BASELINE config 5's command -- `bench.py --gpus 8 --workload mixed --streams 131072` (131,072 mixed 1-64 KiB level-6
    streams per GPU, 1 M on the node) -- with two ranks folded onto device 0 and a smaller per-rank share: the LPT shard
    plan over mixed sizes, every stream compared with its plaintext on each rank."""
    r = _bench_two_ranks(["--workload", "mixed", "--streams", "16384"])
    assert 14000 < r["config"]["streams_per_gpu"] < 19000 and r["config"]["parallelism"] == "shard2"  # (balanced by bytes, not by count)
    assert "config 5" in r["config"]["workload"] and r["value"] > 0
    assert sum(r["per_rank"]["streams"]) == 2 * 16384 and 1.0 <= r["per_rank"]["shard_imbalance"] < 1.05


def test_adler32_over_one_16_gib_device_buffer(gpu_ctx):
    """BASELINE config 2 at its full 16 GiB under -m gpu: one 16 GiB device buffer, SURVEY.md 8d's splitmix64 bytes,
    against chunked zlib.adler32 over the CPU generator's bytes (Adler32.hs:37-57: the combine step multiplies a block's
    byte sum by the length that follows it, a 64-bit number here, modulo 65521).  Also an unaligned start and an odd length."""
    import torch
    dev = torch.device("cuda", 0)
    nwords = (16 << 30) // 8
    buf = torch.empty(nwords, dtype=torch.int64, device=dev)
    chunk = 1 << 25  # words: 256 MiB
    exp, exp_odd = 1, 1
    skew, tail_cut = 5, 3  # the second checksum runs over bytes [5, 16 GiB - 3)
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


def test_streams_beyond_4_gib(gpu_ctx):
    """include/pzg.h: "one compressed stream up to 16 GiB (the reader indexes dwords with 32 bits); the decoded size is not
    limited".  Two streams in one launch, each on one wavefront, rings 11 and 15:
      A  decodes to 4,200 MiB (> 2^32 bytes): 4,200 copies of one full-flushed raw-deflate segment (1 MiB of byte runs and short
         repeating patterns: dist < len copies, far reads, thousands of blocks) between a zlib header and a final empty block;
      B  is 4.3 GB COMPRESSED: 65,600 stored blocks of 65,535 random bytes (level 0), so the bit reader restarts beyond byte 2^32.
    Checked: status, out_len, in_used, the Adler-32 against zlib.adler32 streamed over the plaintext, and EVERY decoded byte on
    the device against the plaintext.  (The bit-at-a-time oracle would need minutes per stream here: the plaintext the streams
    were made from and system zlib's checksum are the references, as for every valid stream.)"""
    import struct
    import torch
    dev = torch.device("cuda", 0)
    # ---- A
    unit = b"".join(corpus.mixed_data(65536, 2 + (k & 1) + 4 * k) for k in range(16))
    assert len(unit) == 1 << 20
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    seg = co.compress(unit) + co.flush(zlib.Z_FULL_FLUSH)  # whole blocks, byte aligned, no reference across the flush
    reps_a = 4200
    ad_a = 1
    for _ in range(reps_a):
        ad_a = zlib.adler32(unit, ad_a)
    za = b"\x78\x9c" + seg * reps_a + b"\x03\x00" + struct.pack(">I", ad_a)
    assert zlib.decompress(b"\x78\x9c" + seg * 2 + b"\x03\x00" + struct.pack(">I", zlib.adler32(unit * 2))) == unit * 2
    len_a = reps_a << 20
    assert len_a > 1 << 32
    # ---- B
    nblk = 65600
    rng = np.random.default_rng(0xB16)
    zb = np.empty(2 + nblk * 65540 + 4, dtype=np.uint8)
    zb[0], zb[1] = 0x78, 0x01
    blocks = zb[2:2 + nblk * 65540].reshape(nblk, 65540)
    blocks[:, 0] = 0
    blocks[-1, 0] = 1  # BFINAL
    blocks[:, 1:3] = 0xFF  # LEN = 65535
    blocks[:, 3:5] = 0x00  # NLEN
    step = 4096
    ad_b = 1
    for lo in range(0, nblk, step):
        hi = min(nblk, lo + step)
        blocks[lo:hi, 5:] = rng.integers(0, 256, size=(hi - lo, 65535), dtype=np.uint8)
        for r in range(lo, hi):
            ad_b = zlib.adler32(blocks[r, 5:], ad_b)
    zb[-4:] = np.frombuffer(struct.pack(">I", ad_b), dtype=np.uint8)
    len_b = nblk * 65535
    assert len(zb) > 1 << 32 and len_b > 1 << 32
    # ---- arenas in HBM
    a_in = (len(za) + 255) // 256 * 256
    d_in = torch.empty(a_in + len(zb) + 256, dtype=torch.uint8, device=dev)
    d_in[:len(za)] = torch.from_numpy(np.frombuffer(za, dtype=np.uint8).copy()).to(dev)
    d_in[a_in:a_in + len(zb)] = torch.from_numpy(zb).to(dev)
    a_out = (len_a + 255) // 256 * 256
    d_out = torch.empty(a_out + len_b + 256, dtype=torch.uint8, device=dev)
    mk = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)  # noqa: E731
    d_in_off, d_in_len = mk([0, a_in]), mk([len(za), len(zb)])
    d_out_off, d_out_cap = mk([0, a_out]), mk([len_a, len_b])
    d_out_len, d_in_used = mk([0, 0]), mk([0, 0])
    d_status = torch.full((2,), -1, dtype=torch.int32, device=dev)
    d_adler = torch.zeros(2, dtype=torch.int32, device=dev)
    d_detail = torch.zeros(4, dtype=torch.int32, device=dev)
    unit_t = torch.from_numpy(np.frombuffer(unit, dtype=np.uint8).copy()).to(dev)
    for ring in (11, 15):
        d_out.fill_(0xCD)
        d_status.fill_(-1)
        torch.cuda.synchronize()
        gpu_ctx.set_ring_bits(ring)
        gpu_ctx.decompress_many_device(d_in.data_ptr(), d_in_off.data_ptr(), d_in_len.data_ptr(), d_out.data_ptr(), d_out_off.data_ptr(),
                                       d_out_cap.data_ptr(), d_out_len.data_ptr(), d_status.data_ptr(), d_detail.data_ptr(),
                                       d_in_used.data_ptr(), d_adler.data_ptr(), 2, sync=True)
        assert d_status.cpu().tolist() == [0, 0], (ring, d_status.cpu().tolist(), d_detail.cpu().tolist())
        assert d_out_len.cpu().tolist() == [len_a, len_b], ring
        assert d_in_used.cpu().tolist() == [len(za), len(zb)], ring
        assert d_adler.cpu().numpy().view(np.uint32).tolist() == [ad_a, ad_b], ring
        got_a = d_out[:len_a].view(reps_a, 1 << 20)
        for lo in range(0, reps_a, 256):
            assert bool((got_a[lo:lo + 256] == unit_t).all()), (ring, lo)
        got_b = d_out[a_out:a_out + len_b].view(nblk, 65535)
        for lo in range(0, nblk, step):
            hi = min(nblk, lo + step)
            assert bool(torch.equal(got_b[lo:hi], torch.from_numpy(blocks[lo:hi, 5:].copy()).to(dev))), (ring, lo)
        assert bool((d_out[len_a:a_out] == 0xCD).all()) and bool((d_out[a_out + len_b:] == 0xCD).all())  # nothing past either extent
    gpu_ctx.set_ring_bits(11)
