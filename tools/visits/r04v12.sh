#!/bin/bash
# round 4, visit 12: wave priorities inside the Jacobi tile kernel (lab: FLUID_XCD_REMAP bits 2 / 3), A/B at 4096^2
OUT=$PWD/gpurun_out/r04v12; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_XCD_REMAP=3" "FLUID_XCD_REMAP=7" "FLUID_XCD_REMAP=11" "FLUID_XCD_REMAP=15" > $OUT/ab_jacobi_setprio.txt 2>&1; cat $OUT/ab_jacobi_setprio.txt
