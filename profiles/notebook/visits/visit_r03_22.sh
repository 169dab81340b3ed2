#!/bin/bash
OUT=$PWD/gpurun_out/r03v22; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "" "FLUID_ADVECT_HEAD=64,1" "FLUID_ADVECT_HEAD=256,1" "FLUID_ADVECT_HEAD=256,2" "FLUID_ADVECT_HEAD=512,2" "FLUID_ADVECT_HEAD=1000,2" 2>&1 | tee $OUT/ab_advect_head.txt
timeout 200 python -m pytest tests/test_hip_vs_oracle.py -m gpu -x -q 2>&1 | tail -2
