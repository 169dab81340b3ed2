#!/bin/bash
# round 6, visit 2: the pressure loop as persistent workgroups taking stacks of tiles from per-XCD ticket heads (k_jacobi_pchain) — first contact:
# (a) bitwise against the per-pass schedule for M = 1 / 2 / 3 tiles per stack and several band heights, (b) A/B against round 5's k_jacobi_tb_chain.
OUT=$PWD/gpurun_out/r06v2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== (a) correctness =="
timeout 1200 python tools/chain_check.py "FLUID_CHAIN_STACK=2" "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=3" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=1" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=3 FLUID_CHAIN_GRID=37" "FLUID_CHAIN_PERSIST=0" 2>&1 | tee $OUT/chain_check.txt
echo "== (b) A/B at 4096^2 / 50 =="
timeout 1500 python tools/ab_env.py --rounds 2 --args "--steps 100 --warmup 30 --no-profile-pass" \
  "FLUID_CHAIN_PERSIST=0" "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=2" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=1" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=3" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_BAND=4" "FLUID_CHAIN_STACK=3" "FLUID_CHAIN_STACK=3 FLUID_CHAIN_BAND=1" 2>&1 | tee $OUT/pchain_ab.txt
echo "== nobody waits (probe, invalid results) =="
timeout 600 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=2" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=1" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=3" 2>&1 | tee -a $OUT/pchain_ab.txt
