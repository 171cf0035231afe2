#!/bin/bash
# Round-6 profile collection for profiles/: the bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes (separate runs, counters only, the bench's own default pool) for several workloads, SQ instruction counters for the
# headline workload, the workload sweep.  Usage (GPU box): tests/tools/r6_profiles.sh [sections: bench stats traffic sq sweeps]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/r6; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
tag=r06_final
sections=${*:-bench stats traffic sq sweeps}
has() { [[ " $sections " == *" $1 "* ]]; }
if has bench; then python3 $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; fi
if has stats; then
  rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --adler-gib 0 > /tmp/ks.log 2>&1
  cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
  rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --adler-gib 0 --incremental-decoders 0 --workload fixed_4k > /tmp/ks.log 2>&1
  cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_fixed_4k.csv
fi
pmc() {  # <label> <counters> <kernel filter> <bench args...>
  label=$1; ctr=$2; filt=$3; shift 3
  rm -rf /tmp/pm; timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --incremental-decoders 0 --adler-gib 0 --steps 3 --warmup 1 "$@" > /tmp/pm.log 2>&1
  python3 - "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" "$label" "$filt" >> $out/${tag}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        acc[(r["Kernel_Name"].split("(")[0][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
for (k, c), v in sorted(acc.items()):
    if sys.argv[3] in k.replace("(int)", ""):
        print(f"{sys.argv[2]:34s} {k[:40]:40s} {c:22s} n={len(v):2d} mean={sum(v)/len(v):.6g}")
PY
}
if has traffic; then
  : > $out/${tag}_pmc.txt
  for wl in ${TRAFFIC_WL:-l6_32k skewed_bytes fixed_4k html mixed}; do
    filt="inflate_kernel<11"; [ $wl = fixed_4k ] && filt="bundle_kernel"   # (config 3's streams are decoded by the bundles' kernel)
    pmc "$wl FETCH(KB)" FETCH_SIZE "$filt" --workload $wl
    pmc "$wl WRITE(KB)" WRITE_SIZE "$filt" --workload $wl
  done
fi
if has sq; then
  has traffic || : > $out/${tag}_pmc.txt
  pmc "l6_32k SQ1" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "inflate_kernel<11" --workload l6_32k
  pmc "l6_32k SQ2" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "inflate_kernel<11" --workload l6_32k
  pmc "l6_32k TCP" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "inflate_kernel<11" --workload l6_32k
  pmc "fixed_4k SQ1" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "bundle_kernel" --workload fixed_4k
  pmc "fixed_4k SQ2" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "bundle_kernel" --workload fixed_4k
fi
if has sweeps; then SWEEP_RINGS=1 $root/tests/tools/sweeps.sh $out/${tag}_sweeps.txt; fi
ls -la $out
