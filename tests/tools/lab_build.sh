#!/bin/bash
# Builds the library with extra -D options (and the product's own compiler flags) into build/lab_<tag>/libpzg.so.  Test
# infrastructure (tests/test_exotic_streams.py, tests/test_gpu_parity.py: the strips with their guesses made to fail); never shipped.
# Usage: tests/tools/lab_build.sh <tag> -DPZG_STRIP_BACK=8 -DPZG_STRIP_ROUNDS=2 ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../../pure_zlib_amd/csrc"
out=../../build/lab_$tag
mkdir -p $out
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -fno-unroll-loops"
/opt/rocm/bin/hipcc $F -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 "$@" -c pzg_kernels.hip -o $out/k.o
/opt/rocm/bin/hipcc $F -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 -mllvm -amdgpu-sdwa-peephole=0 "$@" -c pzg_kernels_b.hip -o $out/kb.o
/opt/rocm/bin/hipcc $F "$@" -c pzg_api.cpp -o $out/api.o
g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -c pzg_errors.cpp -o $out/errors.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $out/k.o $out/kb.o $out/api.o $out/errors.o -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=pzg.map -o $out/libpzg.so
