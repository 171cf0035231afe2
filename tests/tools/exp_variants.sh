#!/bin/bash
# Diagnostic: build several experimental variants, each "tag:flags" -> build/exp/libpzg_<tag>.so (never shipped).
# Usage: tests/tools/exp_variants.sh "base:" "v20:-DPZG_EXP_VALU=20" ...
set -e
cd "$(dirname "$0")/../../pure_zlib_amd/csrc"
mkdir -p ../../build/exp
for tv in "$@"; do
  tag=${tv%%:*}; flags=${tv#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 $flags -c pzg_kernels.hip -o ../../build/exp/k_$tag.o &&
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops $flags -c pzg_api.cpp -o ../../build/exp/a_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../../build/exp/k_$tag.o ../../build/exp/a_$tag.o ../../build/pzg/pzg_errors.o -Wl,-rpath,/opt/rocm/lib -o ../../build/exp/libpzg_$tag.so &&
    rm -f ../../build/exp/k_$tag.o ../../build/exp/a_$tag.o && echo built $tag ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait
