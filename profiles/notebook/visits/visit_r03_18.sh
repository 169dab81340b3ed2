#!/bin/bash
OUT=$PWD/gpurun_out/r03v18; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for N in 2048 3072; do
  echo "== $N ==" | tee -a $OUT/ab_mix_mid_sizes.txt
  timeout 600 python tools/ab_env.py --rounds 2 --args "--size $N --steps 600 --warmup 100 --no-parity" "" "FLUID_TB_VARIANT=0" "FLUID_TB_VARIANT=0 FLUID_FOLD_GRADSUB=0" "FLUID_TB_VARIANT=0 FLUID_TB_TAIL=0,0,7" "FLUID_TB_VARIANT=0 FLUID_TB_TAIL=300,600,5" "FLUID_TB_VARIANT=8" 2>&1 | tee -a $OUT/ab_mix_mid_sizes.txt
done
