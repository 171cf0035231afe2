#!/bin/bash
# Diagnostic: SQ instruction counters of inflate_kernel<11> for one workload (per launch means).  Usage: sq_counters.sh [workload]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
wl=${1:-l6_32k}
cd /tmp; export TMPDIR=/tmp
if [ -n "$PZG_CTRS" ]; then groups=("$PZG_CTRS"); else groups=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"); fi
for ctr in "${groups[@]}"; do
  rm -rf /tmp/pm; timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --adler-gib 0 --pool 2048 --steps 3 --warmup 1 --workload $wl > /tmp/pm.log 2>&1
  python3 - "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" "$wl" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "inflate_kernel<11" in k or "inflate_kernel<(int)11" in k:
        print(f"{sys.argv[2]:14s} {c:22s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
PY
done
