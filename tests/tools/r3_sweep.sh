#!/bin/bash
# Round-3 sweep: bench lines (device-resident, no CPU legs) for the workloads VERDICT r2 names.  Usage: r3_sweep.sh <tag>
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out; mkdir -p $out
tag=$1
: > $out/${tag}_sweeps.txt
line() { python3 $root/bench.py --steps 8 --warmup 2 --no-host-path --cpu-sample 0 --adler-gib 0 --no-ab "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(' '.join(sys.argv[1:]), '->', d['value'], 'GiB/s kernel_ms', d['roofline']['kernel_ms_avg'], 'bit_exact', d['bit_exact'])" "$@" >> $out/${tag}_sweeps.txt; }
line --workload l6_32k
line --workload fixed_4k
line --workload skewed_bytes
line --workload l6_32k --streams 1048576 --blob-bytes 2048 --pool 4096
line --workload mixed --streams 131072
line --workload html
line --workload l6_32k --gzip
cat $out/${tag}_sweeps.txt
