#!/bin/bash
# Round-3 profile collection for profiles/: bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes (separate runs, counters only) for several workloads, SQ instruction counters for the headline workload.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
tag=r03_final
python3 $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
: > $out/${tag}_pmc.txt
pmc() {  # <label> <counters> <bench args...>
  label=$1; ctr=$2; shift 2
  rm -rf /tmp/pm; timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --adler-gib 0 --steps 3 --warmup 1 "$@" > /tmp/pm.log 2>&1
  python3 - "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" "$label" >> $out/${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        acc[(r["Kernel_Name"].split("(")[0][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
for (k, c), v in sorted(acc.items()):
    if "inflate_kernel<11" in k or "inflate_kernel<(int)11" in k:
        print(f"{sys.argv[2]:34s} {c:22s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
PY
}
for wl in l6_32k fixed_4k mixed html skewed_bytes; do
  pmc "$wl FETCH(KB)" FETCH_SIZE --workload $wl
  pmc "$wl WRITE(KB)" WRITE_SIZE --workload $wl
done
pmc "mixed 131072 FETCH(KB)" FETCH_SIZE --workload mixed --streams 131072
pmc "mixed 131072 WRITE(KB)" WRITE_SIZE --workload mixed --streams 131072
pmc "l6_32k SQ1" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --workload l6_32k
pmc "l6_32k SQ2" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" --workload l6_32k
# other workloads, ring classes, residency: bench lines only
: > $out/${tag}_sweeps.txt
line() { python3 $root/bench.py --steps 8 --warmup 2 --no-host-path --cpu-sample 0 --adler-gib 0 --no-ab "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), '->', d['value'], 'GiB/s kernel_ms', d['roofline']['kernel_ms_avg'], 'bit_exact', d['bit_exact'])" "$@" >> $out/${tag}_sweeps.txt; }
for wl in fixed_4k mixed html skewed_bytes fixed_bin runs; do line --workload $wl; done
line --workload mixed --streams 131072
line --workload l6_32k --gzip
line --workload l6_32k --streams 32768 --blob-bytes 65536
line --workload l6_32k --streams 1048576 --blob-bytes 2048 --pool 4096
for rb in 12 13 14 15; do line --ring-bits $rb; done
for w in 2048 4096 5120 6144 6656; do PZG_WAVES=$w line --workload l6_32k; echo "   (PZG_WAVES=$w)" >> $out/${tag}_sweeps.txt; done
cat $out/${tag}_pmc.txt $out/${tag}_sweeps.txt; head -5 $out/${tag}_kernel_stats.csv; cat $out/${tag}_bench.json
