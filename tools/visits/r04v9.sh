#!/bin/bash
# round 4, visit 9: the packed-dye advection with stacked waves and XCD-contiguous block columns (lab knobs): parity at 4096^2, then the A/B
OUT=$PWD/gpurun_out/r04v9; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_properties.py -m gpu -q -x -k "chained_launch_at_the_bench_size or packed_dye" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 200 --warmup 50 --no-parity" "FLUID_SKIP_CURL=1" "FLUID_ADVECT_XCD=1" "FLUID_ADVECT_WY=2 FLUID_ADVECT_XCD=1" "FLUID_ADVECT_WY=4 FLUID_ADVECT_XCD=1" "FLUID_ADVECT_WY=4" "FLUID_ADVECT_WY=4 FLUID_ADVECT_XCD=1 FLUID_ADVECT_ROWS=2" "FLUID_ADVECT_WY=2 FLUID_ADVECT_XCD=1 FLUID_ADVECT_ROWS=2" > $OUT/ab_rgb_wy_xcd_4096.txt 2>&1; cat $OUT/ab_rgb_wy_xcd_4096.txt
