"""Lab tool: profiles/traffic.json from a PMC summary of tests/tools/r6_profiles.sh (lines "<workload> FETCH(KB) ... mean=<KB>").
Every entry records the SHA-256 of the kernels' source and the pool size it was measured with: bench.py prints `traffic: null`
with the reason when either differs from the build it runs.  Usage: python tests/tools/traffic_update.py profiles/r05_final_pmc.txt [pool]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    src = sys.argv[1]
    pool = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_hash", os.path.join(ROOT, "bench.py"))
    # (only the hash function is needed: read it without importing torch)
    import hashlib
    h = hashlib.sha256()
    for fn in ("inflate_core.h", "bundle_core.h", "pzg_inflate_kernel.h", "pzg_bundle_kernel.h", "pzg_kernels.hip", "pzg_kernels_b.hip", "wave.h"):
        with open(os.path.join(ROOT, "pure_zlib_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    sha = h.hexdigest()
    vals = {}
    for ln in open(os.path.join(ROOT, src) if not os.path.isabs(src) else src):
        m = re.match(r"(\S+)(?: (\d+))? (FETCH|WRITE)\(KB\)\s+.*(?:inflate_kernel<11|bundle_kernel).*?(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+ mean=([0-9.e+]+)", ln)
        if m:
            vals.setdefault((m.group(1), int(m.group(2) or 65536)), {})[m.group(3)] = float(m.group(5)) * 1024.0
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    keep = [e for e in t["entries"] if e.get("kernel_sha256") == sha]
    for (wl, streams), v in sorted(vals.items()):
        if "FETCH" in v and "WRITE" in v:
            keep = [e for e in keep if not (e["workload"] == wl and e["streams"] == streams and e["ring_bits"] == 11)]
            keep.append({"workload": wl, "ring_bits": 11, "streams": streams, "gzip": False, "pool": pool, "kernel_sha256": sha,
                         "hbm_bytes_per_launch": int(v["FETCH"] + v["WRITE"]), "fetch_bytes": int(v["FETCH"]), "write_bytes": int(v["WRITE"]),
                         "file": src})
    t["entries"] = keep
    t["source"] = (f"{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of tests/tools/r6_profiles.sh (one pass per counter, counters only; bench.py "
                   f"--steps 3 --warmup 1, its default pool {pool}), mean over the dispatches of the workload's dominant kernel (inflate_kernel<11,false,false>; config 3: bundle_kernel)")
    json.dump(t, open(path, "w"), indent=1)
    print(f"{len(keep)} entries for kernel source {sha[:12]}...")


if __name__ == "__main__":
    main()
