#!/bin/bash
# Lab: bench lines (kernel GiB/s) of the product library and of experimental builds build/exp/libpzg_<tag>.so, interleaved.
# Usage: tests/tools/strip_ab.sh "<tag> ..." [workloads] [passes]
wls=${2:-"l6_32k fixed_4k"}
for pass in $(seq 1 ${3:-2}); do
  for tag in prod $1; do
    lib=$PWD/build/exp/libpzg_$tag.so; [ $tag = prod ] && lib=$PWD/pure_zlib_amd/libpzg.so
    for wl in $wls; do
      PZG_LIB=$lib timeout 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-ab --no-host-path --no-variants --cpu-sample 0 --adler-gib 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', '$wl', d['value'], d['bit_exact'])"
    done
  done
done
