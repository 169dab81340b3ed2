#!/bin/bash
OUT=$PWD/gpurun_out/r03v12; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "knob" 2>&1 | tail -2 | tee $OUT/log.txt
timeout 1200 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_TB_TAIL=0,0,7" "FLUID_TB_TAIL=0,666,7" "FLUID_TB_TAIL=0,426,7" "FLUID_TB_TAIL=0,906,7" "FLUID_TB_TAIL=0,546,6" "FLUID_TB_TAIL=0,306,5" "FLUID_TB_TAIL=186,666,7" "FLUID_TB_TAIL=366,666,7" "FLUID_TB_TAIL=186,426,7" "FLUID_TB_TAIL=200,300,5" 2>&1 | tee $OUT/ab_tail_4096.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 8192 --steps 60 --warmup 10 --no-parity" "FLUID_TB_TAIL=0,0,7" "FLUID_TB_TAIL=0,666,7" "FLUID_TB_TAIL=0,1266,7" 2>&1 | tee $OUT/ab_tail_8192.txt
