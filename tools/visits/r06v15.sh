#!/bin/bash
# round 6, visit 15: the chained launch with BALANCED bands (every XCD the same number of tile rows): bitwise, then the loop map again
OUT=$PWD/gpurun_out/r06v15; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/chain_check.py "" 2>&1 | tee $OUT/chain_check.txt
timeout 1500 python tools/chain_check.py --shapes "8192x2048x50 6144x2730x47 16384x1024x50 2048x8192x33 3072x5460x50 8192x8192x50 16384x2048x200 5000x3000x33" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee -a $OUT/chain_check.txt
timeout 1500 python tools/bench_loop.py --rounds 2 --shapes "4096x4096x50 8192x2048x50 16384x1024x50 2048x8192x50 6144x2730x50 3072x5460x50 5120x3276x50 4096x8192x50 8192x8192x50" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee $OUT/loop_map.txt
