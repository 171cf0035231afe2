run() { python bench.py --steps 6 --warmup 2 --no-host-path --cpu-sample 0 --adler-gib 0 --no-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['bit_exact'])"; }
PZG_WAVES=6656 run full
M5=$(python -c "print('0x'+'5'*64)"); MA=$(python -c "print('0x'+'a'*64)"); M3=$(python -c "print('0x'+'3'*64)"); MF=$(python -c "print('0x'+'0f'*32)")
ROC_GLOBAL_CU_MASK=$M5 PZG_WAVES=3328 run mask5555_w3328
ROC_GLOBAL_CU_MASK=$M3 PZG_WAVES=3328 run mask3333_w3328
ROC_GLOBAL_CU_MASK=$MF PZG_WAVES=3328 run mask0f0f_w3328
HSA_CU_MASK=0:$M5 PZG_WAVES=3328 run hsa5555_w3328
PZG_WAVES=3328 run nomask_w3328
