#!/bin/bash
# Lab: a counter group for the product library and experimental builds.  Usage: strip_ctr.sh "<counters>" "<tag> ..." [workload]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for tag in prod $2; do
  lib=$root/build/exp/libpzg_$tag.so; [ $tag = prod ] && lib=$root/pure_zlib_amd/libpzg.so
  echo "== $tag"; PZG_LIB=$lib PZG_CTRS="$1" timeout 300 bash $root/tests/tools/sq_counters.sh ${3:-l6_32k} 2>&1 | tail -6
done
