#!/bin/bash
# Lab tool: like lab_build.sh, but only pzg_kernels.hip is compiled with the extra options -- the other objects are the product's
# (build/pzg/*.o: run `make` first).  For experiments that live in the zlib kernels / the bundles alone.  -> build/exp/<tag>.so
# Usage: tests/tools/lab_k.sh <tag> -DPZG_BUNDLE_FAR_LAND=1 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../../pure_zlib_amd/csrc"
mkdir -p ../../build/exp ../../build/ab
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -fno-unroll-loops"
/opt/rocm/bin/hipcc $F -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 "$@" -c pzg_kernels.hip -o ../../build/exp/$tag.k.o 2>&1 | grep -E "error" -A5 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../../build/exp/$tag.k.o ../../build/pzg/pzg_kernels_b.o ../../build/pzg/pzg_api.o ../../build/pzg/pzg_errors.o -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=pzg.map -o ../../build/exp/$tag.so
rm -f ../../build/exp/$tag.k.o
cp ../../build/exp/$tag.so ../../build/ab/$tag.so  # (build/exp does not travel to the GPU box, build/ab does)
ls -la ../../build/ab/$tag.so
