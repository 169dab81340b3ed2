#!/bin/bash
OUT=$PWD/gpurun_out/r03v15; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "knob" 2>&1 | tail -2 | tee $OUT/log.txt
timeout 1200 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_CVD_TAIL=0,0" "FLUID_CVD_TAIL=0,252" "FLUID_CVD_TAIL=126,252" "FLUID_CVD_TAIL=252,504" "FLUID_CVD_TAIL=126,126" "FLUID_CVD_TAIL=378,378" 2>&1 | tee $OUT/ab_cvd_tail_4096.txt
