#!/bin/bash
# Diagnostic: bench throughput against the number of resident stream-waves (PZG_WAVES knob).
rb=${1:-11}; shift
for w in "$@"; do
  PZG_WAVES=$w timeout 200 python bench.py --steps 8 --warmup 2 --no-ab --cpu-sample 0 --adler-gib 0 --ring-bits $rb 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves', $w, 'GiB/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms_avg'])"
done
