#!/bin/bash
# Diagnostic: hardware-counter passes over the bench command (one rocprofv3 run per counter group,
# counters only -- never combined with trace domains).  Usage: tests/tools/pmc.sh <ring_bits> "<group1>" "<group2>" ...
# Writes gpurun_out/pmc_summary.txt (per kernel, per counter: mean over dispatches).
rb=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp; export TMPDIR=/tmp
out=$root/gpurun_out; mkdir -p $out; : > $out/pmc_summary.txt
i=0
for grp in "$@"; do
  i=$((i+1)); d=/tmp/pmc_$i; rm -rf $d
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-ab --no-host-path --no-variants --cpu-sample 0 --adler-gib 0 --ring-bits $rb > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "group $grp: no csv"; grep -v simple_timer $d.log | tail -8; find $d | head; continue; fi >> $out/pmc_summary.txt
  python3 - "$f" "$grp" >> $out/pmc_summary.txt <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
except Exception as e:
    print("group", grp, "failed:", e)
for (k, c), v in sorted(acc.items()):
    if "inflate" in k or "adler" in k:
        print(f"{k:40s} {c:28s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
PY
done
cat $out/pmc_summary.txt
