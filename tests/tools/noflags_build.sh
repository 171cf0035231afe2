#!/bin/bash
# Builds the library WITHOUT the two -mllvm options of csrc/Makefile's KERNELFLAGS (the structurizer then rebuilds the hot
# loop's control flow; slower, must be just as correct) into build/noflags/libpzg.so.  Test infrastructure for
# tests/test_abi.py / tests/test_gpu_api.py; never shipped.
set -e
cd "$(dirname "$0")/../../pure_zlib_amd/csrc"
mkdir -p ../../build/noflags
F="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -fno-unroll-loops"
/opt/rocm/bin/hipcc $F -c pzg_kernels.hip -o ../../build/noflags/k.o
/opt/rocm/bin/hipcc $F -c pzg_kernels_b.hip -o ../../build/noflags/kb.o
# (its own API objects, always rebuilt: a stale build/pzg/pzg_api.o would test old host code)
/opt/rocm/bin/hipcc $F -c pzg_api.cpp -o ../../build/noflags/api.o
g++ -O2 -std=c++17 -fPIC -fvisibility=hidden -c pzg_errors.cpp -o ../../build/noflags/errors.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../../build/noflags/k.o ../../build/noflags/kb.o ../../build/noflags/api.o ../../build/noflags/errors.o -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=pzg.map -o ../../build/noflags/libpzg.so
