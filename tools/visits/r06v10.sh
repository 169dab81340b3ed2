#!/bin/bash
# round 6, visit 10: the chained launch's divergence loads in front of the poll (their round trip is the poll's)
OUT=$PWD/gpurun_out/r06v10; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 600 python tools/chain_check.py "" "FLUID_CHAIN_DFIRST=1" 2>&1 | tee $OUT/chain_check.txt
timeout 900 python tools/ab_env.py --rounds 4 --args "--steps 100 --warmup 30 --no-profile-pass" "FLUID_CHAIN_DFIRST=0" "FLUID_CHAIN_DFIRST=1" 2>&1 | tee $OUT/dfirst_ab.txt
