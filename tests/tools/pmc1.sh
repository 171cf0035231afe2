#!/bin/bash
# Lab tool: one rocprofv3 --pmc pass over a kernel-only bench.py command; prints the mean of every counter per kernel.
# Usage (GPU box): tests/tools/pmc1.sh "<counters>" <kernel filter> [bench args...]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
ctr=$1; filt=$2; shift 2
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm; timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --incremental-decoders 0 --adler-gib 0 --steps 3 --warmup 1 "$@" > /tmp/pm.log 2>&1
python3 - "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" "$filt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if sys.argv[2] in k:
        print(f"{k[:44]:44s} {c:26s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
PY
