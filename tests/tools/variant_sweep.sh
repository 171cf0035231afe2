#!/bin/bash
# Diagnostic: bench several experimental builds (build/exp/libpzg_<tag>.so) at several residencies.
# Usage: tests/tools/variant_sweep.sh "<tag>:<waves> ..." [workloads]
wls=${2:-"l6_32k fixed_4k"}
for tv in $1; do
  tag=${tv%%:*}; w=${tv##*:}
  for wl in $wls; do
    PZG_LIB=$PWD/build/exp/libpzg_$tag.so PZG_WAVES=$w timeout 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-ab --no-host-path --no-variants --cpu-sample 0 --adler-gib 0 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', 'waves', $w, '$wl', d['value'], d['bit_exact'])"
  done
done
