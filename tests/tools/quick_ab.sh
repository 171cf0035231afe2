#!/bin/bash
# Lab tool: the kernel-only bench line of a few workloads for the product library and for any number of experimental
# builds (build/exp/*.so given as arguments), interleaved.  Usage (on the GPU box): tests/tools/quick_ab.sh [lib ...]
cd "$(dirname "$0")/../.."
WL=${WL:-"l6_32k html skewed_bytes fixed_4k"}
for rep in ${REPS:-1 2}; do
for w in $WL; do
  for lib in product "$@"; do
    if [ "$lib" = product ]; then unset PZG_LIB; else export PZG_LIB=$lib; fi
    timeout 200 python bench.py --workload $w --no-ab --no-host-path --no-variants --adler-gib 0 --cpu-sample 0 --incremental-decoders 0 --steps 5 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$w', '$lib', d['value'], d['ms_per_step'])"
  done
done
done
