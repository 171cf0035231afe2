#!/bin/bash
# Round-2 first GPU call: the full-size parity tests, a baseline bench line and the per-phase cycle breakdown.
out=gpurun_out; mkdir -p $out
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $out/r2_fullsize.log 2>&1; echo "fullsize rc=$?" >> $out/r2_fullsize.log
python bench.py --steps 10 --warmup 2 --no-host-path --cpu-sample 0 --adler-gib 0 --no-ab > $out/r2_base_bench.json 2> $out/r2_base_bench.err
python tests/tools/prof_run.py 8192 32768 > $out/r2_base_prof.txt 2>&1
tail -5 $out/r2_fullsize.log; cat $out/r2_base_bench.json; cat $out/r2_base_prof.txt
