#!/bin/bash
# Round-4 profile collection for profiles/: the bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE
# passes (separate runs, counters only) for several workloads, the Adler kernel's FETCH_SIZE against its exact byte count (the
# calibration of the fetch correction), SQ instruction / activity counters and the clock (GRBM_GUI_ACTIVE) for the headline workload.
# Usage: tests/tools/r4_profiles.sh [sections: bench stats traffic sq sweeps]   (default: all)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/r4; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
tag=r04_final
sections=${*:-bench stats traffic sq sweeps}
has() { [[ " $sections " == *" $1 "* ]]; }
if has bench; then python3 $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; fi
if has stats; then
  rm -rf /tmp/ks; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants > /tmp/ks.log 2>&1
  cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
fi
pmc() {  # <label> <counters> <kernel filter> <bench args...>
  label=$1; ctr=$2; filt=$3; shift 3
  rm -rf /tmp/pm; timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o p -- python3 $root/bench.py --cpu-sample 0 --no-ab --no-host-path --no-variants --steps 3 --warmup 1 --pool 2048 "$@" > /tmp/pm.log 2>&1
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
  for wl in l6_32k fixed_4k mixed html skewed_bytes; do
    pmc "$wl FETCH(KB)" FETCH_SIZE "inflate_kernel<11" --workload $wl --adler-gib 0
    pmc "$wl WRITE(KB)" WRITE_SIZE "inflate_kernel<11" --workload $wl --adler-gib 0
  done
  pmc "mixed 131072 FETCH(KB)" FETCH_SIZE "inflate_kernel<11" --workload mixed --streams 131072 --adler-gib 0
  pmc "mixed 131072 WRITE(KB)" WRITE_SIZE "inflate_kernel<11" --workload mixed --streams 131072 --adler-gib 0
  # calibration: adler32_partial_kernel reads exactly 4 GiB per launch (16 B per lane, streaming)
  pmc "adler 4 GiB FETCH(KB)" FETCH_SIZE "adler32_partial" --workload l6_32k --streams 4096 --adler-gib 4
fi
if has sq; then
  has traffic || : > $out/${tag}_pmc.txt
  pmc "l6_32k SQ1" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "inflate_kernel<11" --workload l6_32k --adler-gib 0
  pmc "l6_32k SQ2" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" "inflate_kernel<11" --workload l6_32k --adler-gib 0
  pmc "l6_32k SQ3" "SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES" "inflate_kernel<11" --workload l6_32k --adler-gib 0
  pmc "l6_32k GRBM" "GRBM_GUI_ACTIVE GRBM_COUNT" "inflate_kernel<11" --workload l6_32k --adler-gib 0 --steps 10
  pmc "fixed_4k SQ1" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "inflate_kernel<11" --workload fixed_4k --adler-gib 0
fi
if has sweeps; then
  : > $out/${tag}_sweeps.txt
  line() { python3 $root/bench.py --steps 8 --warmup 2 --no-host-path --no-variants --cpu-sample 0 --adler-gib 0 --no-ab "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), '->', d['value'], 'GiB/s kernel_ms', d['roofline']['kernel_ms_avg'], 'bit_exact', d['bit_exact'])" "$@" >> $out/${tag}_sweeps.txt; }
  for wl in fixed_4k mixed html skewed_bytes fixed_bin runs; do line --workload $wl; done
  line --workload mixed --streams 131072
  line --workload l6_32k --gzip
  line --workload l6_32k --streams 32768 --blob-bytes 65536
  line --workload l6_32k --streams 1048576 --blob-bytes 2048 --pool 4096
  for rb in 12 13 14 15; do line --ring-bits $rb; done
fi
cat $out/${tag}_pmc.txt $out/${tag}_sweeps.txt 2>/dev/null; head -5 $out/${tag}_kernel_stats.csv 2>/dev/null
