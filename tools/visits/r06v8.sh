#!/bin/bash
# round 6, visit 8: what bounds the chained Jacobi launch — timing probes (results NOT valid): no arithmetic, no mailbox / barrier, nobody waits
OUT=$PWD/gpurun_out/r06v8; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" \
  "FLUID_CHAIN_SKIP=0" "FLUID_CHAIN_SKIP=1" "FLUID_JACOBI_CHAIN=5" "FLUID_JACOBI_CHAIN=6" "FLUID_JACOBI_CHAIN=4" "FLUID_JACOBI_CHAIN=2" 2>&1 | tee $OUT/chain_bounds.txt
