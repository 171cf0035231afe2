#!/bin/bash
# Diagnostic: FETCH_SIZE / WRITE_SIZE (KB per launch) of inflate_kernel<11> for experimental builds.  Usage: r3_fetch.sh "<tag> ..." [workload]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
wl=${2:-l6_32k}
cd /tmp; export TMPDIR=/tmp
for tag in $1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm; PZG_LIB=$root/build/exp/libpzg_$tag.so timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --adler-gib 0 --steps 3 --warmup 1 --workload $wl > /tmp/pm.log 2>&1
    python3 - "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" "$tag $wl" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "inflate_kernel<11" in k or "inflate_kernel<(int)11" in k:
        print(f"{sys.argv[2]:20s} {c:12s} n={len(v):2d} mean={sum(v)/len(v):.6g} KB")
PY
  done
done
