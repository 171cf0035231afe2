"""Test helper: one pzg_decompress_many launch over arenas resident in HBM (PZG_DEVICE_PTRS), the way
bench.py's timed region calls it.  A batch replicates a pool of distinct (text, stream) pairs at distinct,
256-byte aligned addresses; every stream's status / length / in_used / Adler-32 comes back and every decoded
byte is compared with the expected text."""
import zlib

import numpy as np


class DeviceBatch:
    def __init__(self, texts, zs, pick, dev=0):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", dev)
        self.texts, self.zs = texts, zs
        self.pick = np.asarray(pick, dtype=np.int64)
        n = self.n = len(self.pick)
        zlen = np.array([len(z) for z in zs], dtype=np.int64)
        dlen = np.array([len(t) for t in texts], dtype=np.int64)
        self.in_len = zlen[self.pick]
        self.out_cap = dlen[self.pick]
        self.in_off = np.zeros(n, dtype=np.int64)
        self.out_off = np.zeros(n, dtype=np.int64)
        self.in_off[1:] = np.cumsum((self.in_len[:-1] + 255) // 256 * 256)
        self.out_off[1:] = np.cumsum((self.out_cap[:-1] + 255) // 256 * 256)
        in_bytes = int(self.in_off[-1] + (self.in_len[-1] + 255) // 256 * 256)
        self.out_bytes = int(self.out_off[-1] + (self.out_cap[-1] + 255) // 256 * 256)
        h_in = np.zeros(in_bytes, dtype=np.uint8)
        zarr = [np.frombuffer(z, dtype=np.uint8) for z in zs]
        for k in range(n):
            h_in[self.in_off[k]:self.in_off[k] + self.in_len[k]] = zarr[self.pick[k]]
        self.h_in = h_in
        as_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        self.d_in = as_dev(h_in)
        self.d_out = torch.full((self.out_bytes,), 0xCD, dtype=torch.uint8, device=self.dev)
        self.d_in_off, self.d_in_len = as_dev(self.in_off), as_dev(self.in_len)
        self.d_out_off, self.d_out_cap = as_dev(self.out_off), as_dev(self.out_cap)
        self.d_out_len = torch.zeros(n, dtype=torch.int64, device=self.dev)
        self.d_in_used = torch.zeros(n, dtype=torch.int64, device=self.dev)
        self.d_status = torch.full((n,), -1, dtype=torch.int32, device=self.dev)
        self.d_adler = torch.zeros(n, dtype=torch.int32, device=self.dev)
        self.d_detail = torch.zeros(2 * n, dtype=torch.int32, device=self.dev)
        torch.cuda.synchronize()

    def run(self, ctx, ring_bits):
        t = self.torch
        self.d_out.fill_(0xCD)
        self.d_status.fill_(-1)
        t.cuda.synchronize()
        ctx.set_ring_bits(ring_bits)
        ctx.decompress_many_device(self.d_in.data_ptr(), self.d_in_off.data_ptr(), self.d_in_len.data_ptr(),
                                   self.d_out.data_ptr(), self.d_out_off.data_ptr(), self.d_out_cap.data_ptr(),
                                   self.d_out_len.data_ptr(), self.d_status.data_ptr(), self.d_detail.data_ptr(),
                                   self.d_in_used.data_ptr(), self.d_adler.data_ptr(), self.n, sync=True)
        return (self.d_status.cpu().numpy(), self.d_out_len.cpu().numpy(), self.d_in_used.cpu().numpy(),
                self.d_adler.cpu().numpy().view(np.uint32))

    def check_all(self, status, out_len, in_used, adler):
        """status, length, in_used, Adler-32 of every stream, then every decoded byte."""
        assert (status == 0).all(), np.nonzero(status != 0)[0][:8]
        assert (out_len == self.out_cap).all()
        assert (in_used == self.in_len).all()
        exp = np.array([zlib.adler32(t) for t in self.texts], dtype=np.uint32)[self.pick]
        assert (adler == exp).all()
        t = self.torch
        widths = set(len(x) for x in self.texts)
        if len(widths) == 1:  # equal sizes: compare on the device
            width = widths.pop()
            stride = (width + 255) // 256 * 256
            pool_t = t.from_numpy(np.frombuffer(b"".join(self.texts), dtype=np.uint8).reshape(len(self.texts), width).copy()).to(self.dev)
            got = self.d_out.view(self.n, stride)
            idx = t.from_numpy(self.pick).to(self.dev)
            for lo in range(0, self.n, 8192):
                assert bool(t.equal(got[lo:lo + 8192, :width], pool_t[idx[lo:lo + 8192]])), lo
                if stride > width:  # the gaps between extents stay untouched
                    assert bool((got[lo:lo + 8192, width:] == 0xCD).all()), lo
        else:  # mixed sizes: the arena comes back in slices of ~512 MiB and every extent is compared on the host
            step = 1 << 29
            k = 0
            tarr = [np.frombuffer(x, dtype=np.uint8) for x in self.texts]
            while k < self.n:
                lo = int(self.out_off[k])
                k2 = int(np.searchsorted(self.out_off, lo + step, side="left"))
                k2 = max(k2, k + 1)
                hi = int(self.out_off[k2]) if k2 < self.n else self.out_bytes
                h = self.d_out[lo:hi].cpu().numpy()
                for q in range(k, k2):
                    o = int(self.out_off[q]) - lo
                    assert np.array_equal(h[o:o + int(self.out_cap[q])], tarr[self.pick[q]]), q
                k = k2

    def check_sample_vs_oracle(self, oracle, count=256, seed=7):
        """`count` sampled streams: the oracle's bytes / adler / in_used against what the device left in HBM."""
        rng = np.random.default_rng(seed)
        adler = self.d_adler.cpu().numpy().view(np.uint32)
        in_used = self.d_in_used.cpu().numpy()
        for k in rng.choice(self.n, size=min(count, self.n), replace=False):
            z = self.zs[self.pick[k]]
            r, o = oracle.decompress(z, int(self.out_cap[k]))
            lo = int(self.out_off[k])
            got = self.d_out[lo:lo + int(self.out_cap[k])].cpu().numpy().tobytes()
            assert r.status == 0 and got == o, k
            assert int(adler[k]) == r.adler and int(in_used[k]) == r.in_used, k
