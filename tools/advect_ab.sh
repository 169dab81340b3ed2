#!/bin/bash
# GPU-box A/B of the advection kernel: general (FLUID_ADVECT_FAST=0) against the fast variant, pass time at 4096^2 and whole-step rate.
OUT=gpurun_out/${1:-advect_ab}; mkdir -p $OUT
for fr in "0 2" "1 1" "1 2" "1 4"; do
  set -- $fr
  echo "== FLUID_ADVECT_FAST=$1 ROWS=$2 ==" | tee -a $OUT/log.txt
  FLUID_ADVECT_FAST=$1 FLUID_ADVECT_ROWS=$2 python tools/bench_pass.py 4096 2>&1 | tail -1 | cut -c1-80 | tee -a $OUT/log.txt
done
for fr in "0 2" "1 2" "1 4"; do
  set -- $fr
  echo "== bench FLUID_ADVECT_FAST=$1 ROWS=$2 ==" | tee -a $OUT/log.txt
  FLUID_ADVECT_FAST=$1 FLUID_ADVECT_ROWS=$2 python bench.py --cpu-budget 0 --no-traffic --no-steady 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps_per_sec'], d['ms_per_step'], d['pass_ms_per_step'])" | tee -a $OUT/log.txt
done
