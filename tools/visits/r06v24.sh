#!/bin/bash
# round 6, visit 24: repeatability of the rotated chained launch at 3072^2 (one run of visit 23 read 0.358 ms against 0.283), and the stripe rank under rotations 0 / 3 / 5
OUT=$PWD/gpurun_out/r06v24; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 6 --args "--size 3072 --steps 150 --warmup 40 --no-profile-pass --no-parity" "" "FLUID_JACOBI_CHAIN=0" "FLUID_CHAIN_ROT=3" 2>&1 | sed "s/^/[3072] /" | tee $OUT/step_3072.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for k in 1 2 3; do
  for st in "FLUID_CHAIN_ROT=5" "FLUID_CHAIN_ROT=0" "FLUID_CHAIN_ROT=3"; do
    echo "== one rank alone (stripe) [$st] =="
    env FLUID_HIP_LIB=$PROBES $st timeout 600 python tools/overlap_vs_link.py --config stripe --quick --rounds 1 2>&1 | grep "link   0" | tee -a $OUT/rank_stripe.txt
  done
done
