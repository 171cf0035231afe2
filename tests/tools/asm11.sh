#!/bin/bash
# Diagnostic: device assembly of inflate_kernel<11,false,false> -> build/asm/<tag>.s, with its register / LDS budget
# and static instruction counts.  Usage: tests/tools/asm11.sh <tag> [extra flags]
tag=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/build/asm
cd $root/pure_zlib_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-unroll-loops -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -align-all-nofallthru-blocks=5 --cuda-device-only "$@" -S pzg_kernels.hip -o $root/build/asm/${tag}_all.s 2>&1 | grep -v "warning: argument unused"
cd $root/build/asm
awk '/^_ZN3pzg14inflate_kernelILi11ELb0ELb0EEEvNS_11InflateArgsE:/{f=1} f{print} /^\.Lfunc_end/{if(f){exit}}' ${tag}_all.s > $tag.s
awk '/amdhsa_kernel _ZN3pzg14inflate_kernelILi11ELb0ELb0E/{f=1} f && /next_free_vgpr|next_free_sgpr|group_segment_fixed/{print} /end_amdhsa_kernel/{f=0}' ${tag}_all.s
echo "static: VALU $(grep -c '^\s*v_' $tag.s) SALU $(grep -c '^\s*s_' $tag.s) LDS $(grep -c '^\s*ds_' $tag.s)"
