#!/bin/bash
OUT=$PWD/gpurun_out/r03v19; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for N in 4096 3072 8192; do
  ST=200; [ $N = 8192 ] && ST=60; [ $N = 3072 ] && ST=300
  echo "== $N ==" | tee -a $OUT/ab_mix_shapes.txt
  timeout 900 python tools/ab_env.py --rounds 3 --args "--size $N --steps $ST --warmup 40 --no-parity" "FLUID_TB_TAIL=366,666,7" "FLUID_TB_TAIL=300,600,5" "FLUID_TB_TAIL=400,500,5" "FLUID_TB_TAIL=300,600,6" 2>&1 | tee -a $OUT/ab_mix_shapes.txt
done
