"""Lab tool: the headline batch through the host-pointer path, staged and PZG_HOST_PINNED, with the lab library's trace
(PZG_LIB=build/exp/libpzg_lab.so PZG_TRACE_HOST=1).  Prints ms per call."""
import os, sys, time, zlib
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import pure_zlib_amd as P
from pure_zlib_amd.zlib import PinnedArena
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
texts = [corpus.zipf_text(32768, s) for s in range(512)]
zs = [zlib.compress(t, 6) for t in texts]
pick = np.random.default_rng(1).integers(0, len(zs), size=n)
in_len = np.array([len(zs[k]) for k in pick], dtype=np.uint64)
out_cap = np.full(n, 32768, dtype=np.uint64)
in_off = np.concatenate([[0], np.cumsum((in_len[:-1] + 15) // 16 * 16)]).astype(np.uint64)
out_off = (np.arange(n, dtype=np.uint64) * np.uint64(32768))
p_in, p_out = PinnedArena(int(in_off[-1] + in_len[-1]) + 64), PinnedArena(n * 32768 + 64)
for k in range(n):
    p_in.a[int(in_off[k]):int(in_off[k]) + int(in_len[k])] = np.frombuffer(zs[pick[k]], dtype=np.uint8)
h_in, h_out = p_in.a.copy(), np.empty(p_out.nbytes, dtype=np.uint8)
ctx = P.Context(0)
if len(sys.argv) > 2:  # what bench.py does before its host-path legs: device-resident arenas, its own stream, a few device-pointer launches
    as_dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to("cuda")
    d_in = torch.from_numpy(h_in).to("cuda"); d_out = torch.zeros(p_out.nbytes, dtype=torch.uint8, device="cuda")
    d_io, d_il, d_oo, d_oc = as_dev(in_off), as_dev(in_len), as_dev(out_off), as_dev(out_cap)
    d_ol = torch.zeros(n, dtype=torch.int64, device="cuda"); d_us = torch.zeros(n, dtype=torch.int64, device="cuda")
    d_st = torch.zeros(n, dtype=torch.int32, device="cuda"); d_ad = torch.zeros(n, dtype=torch.int32, device="cuda"); d_de = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    if "stream" in sys.argv[2]:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(4):
        ctx.decompress_many_device(d_in.data_ptr(), d_io.data_ptr(), d_il.data_ptr(), d_out.data_ptr(), d_oo.data_ptr(), d_oc.data_ptr(), d_ol.data_ptr(),
                                   d_st.data_ptr(), d_de.data_ptr(), d_us.data_ptr(), d_ad.data_ptr(), n, sync=True)
    print("device path ms", ctx.last_kernel_ms(), flush=True)
    if "r15" in sys.argv[2]:
        ctx.set_ring_bits(15)
        ctx.decompress_many_device(d_in.data_ptr(), d_io.data_ptr(), d_il.data_ptr(), d_out.data_ptr(), d_oo.data_ptr(), d_oc.data_ptr(), d_ol.data_ptr(),
                                   d_st.data_ptr(), d_de.data_ptr(), d_us.data_ptr(), d_ad.data_ptr(), n, sync=True)
        ctx.set_ring_bits(11)
for name, a, b, pin in (("staged", h_in, h_out, False), ("pinned", p_in.a, p_out.a, True), ("staged", h_in, h_out, False), ("pinned", p_in.a, p_out.a, True), ("pinned", p_in.a, p_out.a, True)):
    t0 = time.perf_counter()
    r = ctx.decompress_many_raw(a, in_off, in_len, b, out_off, out_cap, pinned=pin)
    dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.1f} ms  {n * 32768 / dt / 2**30:.1f} GiB/s  ok={bool((r[1] == 0).all())} kernel_ms={ctx.last_kernel_ms():.2f}", flush=True)
# raw copy rates into / out of the pinned arenas for reference
d = torch.empty(p_out.nbytes, dtype=torch.uint8, device="cuda")
t_out = torch.from_numpy(p_out.a)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); t_out.copy_(d); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"torch D2H into the pinned arena: {dt * 1e3:.1f} ms = {p_out.nbytes / dt / 1e9:.1f} GB/s")
