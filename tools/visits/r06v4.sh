#!/bin/bash
# round 6, visit 4: k_jacobi_pchain, second form (the next ticket drawn with the last tile's loads, decoded and polled while the stores drain)
OUT=$PWD/gpurun_out/r06v4; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== correctness =="
timeout 900 python tools/chain_check.py "FLUID_CHAIN_STACK=2" "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=3 FLUID_CHAIN_BAND=1" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=3 FLUID_CHAIN_GRID=37" 2>&1 | tee $OUT/chain_check.txt
echo "== statistics =="
for S in "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=2" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=1" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=2" "FLUID_CHAIN_STACK=3 FLUID_CHAIN_BAND=1"; do
  echo "--- $S" | tee -a $OUT/pchain_stats.txt
  env $S FLUID_CHAIN_STATS=1 FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so timeout 300 python bench.py --steps 100 --warmup 30 --cpu-budget 0 --no-traffic --no-steady --no-profile-pass --no-parity 2>&1 >/dev/null | grep "pchain stats" | tail -1 | tee -a $OUT/pchain_stats.txt
done
echo "== A/B =="
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 100 --warmup 30 --no-profile-pass" \
  "FLUID_CHAIN_PERSIST=0" "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=2" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=1" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=2" "FLUID_CHAIN_STACK=3 FLUID_CHAIN_BAND=1" 2>&1 | tee $OUT/pchain_ab.txt
timeout 600 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=2" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=1" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_PERSIST=0" 2>&1 | tee -a $OUT/pchain_ab.txt
