#!/bin/bash
OUT=$PWD/gpurun_out/r03v11; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "knob" 2>&1 | tail -2 | tee $OUT/log.txt
for t in "0,5" "186,5" "306,5" "426,5" "546,5" "666,5" "306,6" "546,6" "786,6" "426,7" "666,7" "906,7"; do
  echo -n "FLUID_TB_TAIL=$t  " | tee -a $OUT/tail_launch.txt
  FLUID_TB_TAIL=$t _JIC_CHILD=1 FLUID_TB_VARIANT=0 timeout 120 python tools/jacobi_iter_cost.py 4096 2>/dev/null | tail -1 | tee -a $OUT/tail_launch.txt
done
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 200 --warmup 50 --no-parity" "FLUID_TB_TAIL=0,5" "FLUID_TB_TAIL=306,5" "FLUID_TB_TAIL=546,5" "FLUID_TB_TAIL=546,6" "FLUID_TB_TAIL=666,7" 2>&1 | tee $OUT/ab_tail_4096.txt
