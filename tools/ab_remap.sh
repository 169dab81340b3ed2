#!/bin/bash
# On the GPU box: Jacobi tile-order (FLUID_XCD_REMAP 0..3) x variant x iteration-count sweep.
TAG=$1; VARS=$2; ITERS=${3:-"50"}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for IT in $ITERS; do
  for R in 0 1 2 3; do
    echo "=== remap $R iters $IT ===" | tee -a $OUT/remap.txt
    FLUID_XCD_REMAP=$R TB_VARIANTS="$VARS" python tools/bench_jacobi.py 4096 $IT 2>&1 | tee -a $OUT/remap.txt
  done
done
