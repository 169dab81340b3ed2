#!/bin/bash
# round 6, visit 13: (a) the stale-row skipping (FLUID_CHAIN_SKIP=1: first / last wave stop sweeping rows the apron has reached) where the loop is
# VALU-bound — 8192^2, chained launches both; (b) 4096^2 at 200 iterations: 20 chained blocks (the limit was 8) against 20 plain launches
OUT=$PWD/gpurun_out/r06v13; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 2 --args "--size 8192 --steps 40 --warmup 10 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_SKIP=1" 2>&1 | tee $OUT/skip_8192.txt
timeout 900 python tools/chain_check.py --shapes "4096x4096x200 4096x3072x130" "" "FLUID_JACOBI_CHAIN=0" 2>&1 | tee $OUT/chain_check_200.txt
timeout 900 python tools/ab_env.py --rounds 2 --args "--size 4096 --iters 200 --steps 40 --warmup 10 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee $OUT/chain_4096_200.txt
