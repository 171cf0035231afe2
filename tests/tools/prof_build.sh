#!/bin/bash
# Diagnostic build with in-kernel cycle stamps (never shipped): writes build/prof/libpzg.so
set -e
cd "$(dirname "$0")/../../pure_zlib_amd/csrc"
mkdir -p ../../build/prof
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 -DPZG_PROFILE $PZG_EXTRA -c pzg_kernels.hip -o ../../build/prof/k.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 -mllvm -amdgpu-sdwa-peephole=0 -DPZG_PROFILE $PZG_EXTRA -c pzg_kernels_b.hip -o ../../build/prof/kb.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-unroll-loops -DPZG_PROFILE $PZG_EXTRA -c pzg_api.cpp -o ../../build/prof/a.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../../build/prof/k.o ../../build/prof/kb.o ../../build/prof/a.o ../../build/pzg/pzg_errors.o -Wl,-rpath,/opt/rocm/lib -o ../../build/prof/libpzg.so
