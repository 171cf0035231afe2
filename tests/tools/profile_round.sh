#!/bin/bash
# Collects what profiles/ holds for one kernel revision: the bench line, rocprofv3 kernel stats of the same
# command, and the FETCH_SIZE / WRITE_SIZE counter passes (separate runs, counters only).
# Usage: tests/tools/profile_round.sh <tag> [ring_bits]     -> gpurun_out/<tag>_*
tag=$1; rb=${2:-11}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
python3 $root/bench.py --ring-bits $rb > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $root/bench.py --ring-bits $rb --cpu-sample 0 --no-ab --no-host-path > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -o p -- python3 $root/bench.py --ring-bits $rb --cpu-sample 0 --no-ab --no-host-path --adler-gib 4 --steps 3 --warmup 1 > /tmp/pm_$c.log 2>&1
  python3 - $(find /tmp/pm_$c -name "*counter_collection.csv" | head -1) $c >> $out/${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[2]:12s} {k:44s} n={len(v):2d} mean={sum(v)/len(v):.6g} (KB as reported)")
PY
done
cat $out/${tag}_pmc.txt; cat $out/${tag}_kernel_stats.csv | head -8; cat $out/${tag}_bench.json
