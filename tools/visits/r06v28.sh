#!/bin/bash
# round 6, visit 28: grids BELOW 3072^2 on the chained launch (lab: the large-grid tile forced, the gradient subtract not folded, chain wherever it can) against the
# product's small-grid schedule (40-row tiles, five launches, the last one with the gradient subtract)
OUT=$PWD/gpurun_out/r06v28; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for sz in 2048 2560 1536 1024; do
  timeout 600 python tools/ab_env.py --rounds 2 --args "--size $sz --steps 400 --warmup 100 --no-profile-pass --no-parity" "FLUID_CHAIN_ROT=5" "FLUID_TB_SMALL=1:0 FLUID_FOLD_GRADSUB=0 FLUID_JACOBI_CHAIN=1" "FLUID_TB_SMALL=1:0 FLUID_FOLD_GRADSUB=0 FLUID_JACOBI_CHAIN=0" 2>&1 | sed "s/^/[$sz] /" | cut -c1-130 | tee -a $OUT/small_grids_chain.txt
done
