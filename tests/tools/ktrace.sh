#!/bin/bash
# Lab tool: rocprofv3 kernel trace of one kernel-only bench.py command; prints start offsets / durations of the last launches' kernels.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt -o kt -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --incremental-decoders 0 --adler-gib 0 --no-verify --steps 4 "$@" > /tmp/kt.log 2>&1
python3 - "$(find /tmp/kt -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "pzg::" in r["Kernel_Name"]]
t0 = ks[-12][1] if len(ks) >= 12 else ks[0][1]
prev_end = None
for name, s, e in ks[-12:]:
    gap = None if prev_end is None else (s - prev_end) / 1e3
    print(f"{name:40s} start {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  gap_before {gap}")
    prev_end = e
PY
