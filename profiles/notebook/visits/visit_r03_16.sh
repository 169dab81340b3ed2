#!/bin/bash
OUT=$PWD/gpurun_out/r03v16; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_CVD_TAIL=0,0" "FLUID_CVD_TAIL=378,378" "FLUID_CVD_TAIL=378,0" "FLUID_CVD_TAIL=630,252" "FLUID_CVD_TAIL=882,252" "FLUID_CVD_TAIL=1134,126" 2>&1 | tee $OUT/ab_cvd_tail_4096_b.txt
