#!/bin/bash
# Diagnostic: build an experimental libpzg.so with extra -D flags into build/exp/ (never shipped).
# Usage: tests/tools/exp_build.sh -DPZG_NO_FAR_FENCE ...   then  PZG_LIB=build/exp/libpzg.so python bench.py ...
set -e
cd "$(dirname "$0")/../../pure_zlib_amd/csrc"
mkdir -p ../../build/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 "$@" -c pzg_kernels.hip -o ../../build/exp/k.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 -mllvm -amdgpu-sdwa-peephole=0 "$@" -c pzg_kernels_b.hip -o ../../build/exp/kb.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops "$@" -c pzg_api.cpp -o ../../build/exp/a.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../../build/exp/k.o ../../build/exp/kb.o ../../build/exp/a.o ../../build/pzg/pzg_errors.o -Wl,-rpath,/opt/rocm/lib -o ../../build/exp/libpzg.so
