#!/bin/bash
# round 6, visit 20: the chained launch with the XCD <-> band assignment ROTATED from block to block (FLUID_CHAIN_ROT=r: block l gives slot-XCD k
# the bands (k + l r) % 8): does evening out the XCDs' loads over the launch turn the shapes with a partly empty last group around?
OUT=$PWD/gpurun_out/r06v20; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/chain_check.py --shapes "8192x2048x50 6144x2730x47 16384x1024x50 4096x4096x50 3800x2600x80" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=4" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=3" 2>&1 | tee $OUT/chain_check.txt
timeout 1500 python tools/bench_loop.py --rounds 2 --shapes "4096x4096x50 8192x2048x50 16384x1024x50 2048x8192x50 6144x2730x50 3072x5460x50" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=4" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=3" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_ROT=1" 2>&1 | tee $OUT/loop_map_rot.txt
