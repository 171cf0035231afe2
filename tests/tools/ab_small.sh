#!/bin/bash
# Lab tool: small-stream workloads (1 M x 2 KiB, 512 K x 4 KiB level-6; the headline; config 3 without bundles) for the libraries of $LIBS
# ("product" = the shipped one), kernel-only bench lines.  Usage (GPU box): LIBS="product build/ab/x.so" tests/tools/ab_small.sh
cd "$(dirname "$0")/../.."
run() { # lib args...
  lib=$1; shift
  if [ "$lib" = product ]; then unset PZG_LIB; else export PZG_LIB=$lib; fi
  timeout 300 python bench.py "$@" --no-ab --no-host-path --no-variants --adler-gib 0 --cpu-sample 0 --incremental-decoders 0 --steps 5 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$lib', '$*', d['value'], d['ms_per_step'], d['bit_exact'])"
}
for lib in ${LIBS:-product}; do
  run $lib --workload l6_32k --streams 1048576 --blob-bytes 2048 --pool 4096
  run $lib --workload l6_32k --streams 524288 --blob-bytes 4096 --pool 4096
  run $lib --workload l6_32k --streams 262144 --blob-bytes 8192 --pool 4096
  run $lib --workload l6_32k
  run $lib --workload fixed_4k --bundles 0
done
