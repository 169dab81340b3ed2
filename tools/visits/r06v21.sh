#!/bin/bash
# round 6, visit 21: the rotation (FLUID_CHAIN_ROT) on the whole step at the headline shape and at 200 iterations, and the loop map for the rule
OUT=$PWD/gpurun_out/r06v21; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 100 --warmup 30 --no-profile-pass --no-parity" "FLUID_CHAIN_ROT=0" "FLUID_CHAIN_ROT=3" "FLUID_CHAIN_ROT=5" "FLUID_CHAIN_ROT=1" 2>&1 | tee $OUT/rot_step_4096.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--iters 200 --steps 40 --warmup 10 --no-profile-pass --no-parity" "FLUID_CHAIN_ROT=0" "FLUID_CHAIN_ROT=3" 2>&1 | tee $OUT/rot_step_4096_200.txt
timeout 1500 python tools/bench_loop.py --rounds 2 --shapes "3072x3072x50 4096x2560x50 5120x3276x50 12288x1366x50 4096x4880x50 4096x5461x50 4096x6144x50 4096x8192x50 8192x8192x50 16384x2048x200" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=3" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=0" 2>&1 | tee $OUT/loop_map_rot3.txt
