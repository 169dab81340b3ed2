#!/bin/bash
# round 6, visit 22: which rotation — 3, 5 or 7 bands per block
OUT=$PWD/gpurun_out/r06v22; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python tools/bench_loop.py --rounds 3 --shapes "4096x4096x50 8192x2048x50 16384x1024x50 2048x8192x50 6144x2730x50 3072x5460x50 5120x3276x50 4096x2560x50" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=3" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=5" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=7" 2>&1 | tee $OUT/loop_map_rot357.txt
timeout 900 python tools/ab_env.py --rounds 4 --args "--steps 100 --warmup 30 --no-profile-pass --no-parity" "FLUID_CHAIN_ROT=3" "FLUID_CHAIN_ROT=5" "FLUID_CHAIN_ROT=7" 2>&1 | tee $OUT/rot_step_4096.txt
