#!/bin/bash
# per-phase cycle stamps (diagnostic build) + SQ instruction counters of the shipped build
tag=$1
out=gpurun_out; mkdir -p $out
python tests/tools/prof_run.py 8192 32768 > $out/${tag}_prof.txt 2>&1
bash tests/tools/pmc.sh 11 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" > /dev/null 2>&1
cp $out/pmc_summary.txt $out/${tag}_sq.txt
cat $out/${tag}_prof.txt; grep inflate_kernel.11 $out/${tag}_sq.txt
