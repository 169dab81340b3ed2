#!/bin/bash
# round 4, visit 23: rows per thread of the dye != sim advection with the velocity run in LDS, unpacked dye (dye grids below 3072^2)
OUT=$PWD/gpurun_out/r04v23; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for cfg in "512 2048" "128 1024" "256 2816"; do set -- $cfg
timeout 600 python tools/ab_passes.py --sim $1 --dye $2 --iters 20 --rounds 2 "FLUID_ADVECT_SPLIT_ROWS=2" "FLUID_ADVECT_SPLIT_ROWS=1" "FLUID_ADVECT_SPLIT_ROWS=4" 2>&1 | cut -c1-260 | tee -a $OUT/ab_vtile_rows_unpacked.txt
done
