#!/bin/bash
# Lab tool: the kernel-only rate of level-6 text streams by stream size, 2 GiB decoded per launch.  Usage (GPU box): tests/tools/size_curve.sh
cd "$(dirname "$0")/../.."
for kb in 1 2 3 4 6 8 12 16 24 32 64 128; do
  n=$((2097152 / kb))
  timeout 300 python bench.py --workload l6_32k --streams $n --blob-bytes $((kb * 1024)) --pool 2048 --no-ab --no-host-path --no-variants --adler-gib 0 --cpu-sample 0 --incremental-decoders 0 --steps 5 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$kb KiB x $n:', d['value'], 'GiB/s', d['ms_per_step'], 'ms', d['bit_exact'])"
done
