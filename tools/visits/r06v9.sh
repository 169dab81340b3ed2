#!/bin/bash
# round 6, visit 9: k_jacobi_pchain, fourth form (owned positions, claims, the next item prepared in the loads' shadow, the drain behind the next loads)
OUT=$PWD/gpurun_out/r06v9; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== correctness =="
timeout 900 python tools/chain_check.py "FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=1" "FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=2" "FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=1 FLUID_CHAIN_GRID=40" 2>&1 | tee $OUT/chain_check.txt
echo "== A/B =="
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 100 --warmup 30 --no-profile-pass" \
  "FLUID_CHAIN_SKIP=0" "FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=1" "FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=2" "FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=1 FLUID_CHAIN_BAND=1" 2>&1 | tee $OUT/pchain_ab.txt
timeout 600 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=1" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_PERSIST=1 FLUID_CHAIN_STACK=2" "FLUID_JACOBI_CHAIN=4" 2>&1 | tee -a $OUT/pchain_ab.txt
