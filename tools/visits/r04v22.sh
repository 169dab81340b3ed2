#!/bin/bash
# round 4, visit 22: rows per thread of the dye != sim advection with the velocity run in LDS (lab: FLUID_ADVECT_SPLIT_ROWS)
OUT=$PWD/gpurun_out/r04v22; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 2 "FLUID_ADVECT_SPLIT_ROWS=2" "FLUID_ADVECT_SPLIT_ROWS=1" "FLUID_ADVECT_SPLIT_ROWS=4" > $OUT/ab_vtile_rows.txt 2>&1; cat $OUT/ab_vtile_rows.txt | cut -c1-260
