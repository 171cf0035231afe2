#!/bin/bash
# Diagnostic: timeline (memory copies + kernels) of one host-buffer call of the headline batch
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ht; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ht -o ht -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --adler-gib 0 --no-ab --no-verify > /tmp/ht.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/ht/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], int(r.get("Bytes", r.get("Size", 0)) or 0)))
for f in glob.glob("/tmp/ht/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate_kernel<11" in r["Kernel_Name"] or "inflate_kernel<(int)11" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL", 0))
rows.sort()
big = [r for r in rows if r[3] > (8 << 20) or r[2] == "KERNEL"]
big = big[-60:]
t0 = big[0][0]
for s, e, d, b in big:
    print(f"{(s - t0) / 1e6:9.2f} ms  +{(e - s) / 1e6:7.2f} ms  {d:28s} {b / 2**20:8.1f} MiB  {b / max(e - s, 1):6.1f} GB/s")
PY
