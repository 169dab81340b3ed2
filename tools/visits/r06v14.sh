#!/bin/bash
# round 6, visit 14: where does the chained launch pay?  The loop alone over widths (tiles per row) x set sizes (Infinity Cache: 256 MB; the
# loop's set is 12 B/texel: 4096^2 = 201 MB)
OUT=$PWD/gpurun_out/r06v14; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1500 python tools/bench_loop.py --rounds 2 --shapes "4096x4096x50 4096x8192x50 4096x16384x50 8192x2048x50 16384x1024x50 2048x8192x50 8192x8192x50 6144x2730x50 3072x5460x50" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee $OUT/loop_map.txt
