#!/bin/bash
# Quick GPU check of a kernel revision: the core parity tests, then bench lines for the main workloads.
# Usage: tests/tools/quick.sh <tag> [extra workloads...]
tag=$1; shift
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or pinned or valid_streams or corrupt or far_back or long_codes or capacity or unaligned" > $out/${tag}_tests.log 2>&1
tail -3 $out/${tag}_tests.log
for wl in l6_32k "$@"; do
  python bench.py --workload $wl --steps 10 --warmup 2 --no-host-path --cpu-sample 0 --adler-gib 0 --no-ab 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$tag', '$wl', d['value'], 'GiB/s', 'kernel_ms', d['roofline']['kernel_ms_avg'], 'bit_exact', d['bit_exact'])" | tee -a $out/${tag}_bench.txt
done
