#!/bin/bash
# round 6, visit 3: where a persistent workgroup's time goes (lab statistics of k_jacobi_pchain), after the host-memory read left the poll loop
OUT=$PWD/gpurun_out/r06v3; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
for S in "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=2" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=1" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=2" "FLUID_CHAIN_STACK=3 FLUID_CHAIN_BAND=1"; do
  echo "--- $S (statistics build: timestamps in the kernel)"
  env $S FLUID_CHAIN_STATS=1 FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so timeout 300 python bench.py --steps 100 --warmup 30 --cpu-budget 0 --no-traffic --no-steady --no-profile-pass --no-parity 2>&1 >/dev/null | grep "pchain stats" | tail -1 | tee -a $OUT/pchain_stats.txt
done
