#!/bin/bash
# round 6, visit 25: with the rotation in, rows per band and panels once more at 4096^2 (round 5 swept them under the fixed assignment)
OUT=$PWD/gpurun_out/r06v25; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python tools/bench_loop.py --rounds 3 --shapes "4096x4096x50" "FLUID_CHAIN_ROT=5" "FLUID_CHAIN_BAND=2" "FLUID_CHAIN_BAND=1" "FLUID_CHAIN_BAND=4" "FLUID_CHAIN_PANEL=9" "FLUID_CHAIN_PANEL=6" "FLUID_CHAIN_DFIRST=1" "FLUID_CHAIN_SKIP=1" 2>&1 | tee $OUT/band_sweep_4096.txt
