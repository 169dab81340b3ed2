#!/bin/bash
# round 6, visit 5: do persistent workgroups stay in phase?  A start delay of 0 ... S us per workgroup (hash of its number), nobody waiting (probe) and for real
OUT=$PWD/gpurun_out/r06v5; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_PERSIST=0" \
  "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=1" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=1 FLUID_CHAIN_STAGGER_US=8" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=1 FLUID_CHAIN_STAGGER_US=16" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=1 FLUID_CHAIN_STAGGER_US=30" \
  "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=2" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=2 FLUID_CHAIN_STAGGER_US=16" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_STACK=2 FLUID_CHAIN_STAGGER_US=30" 2>&1 | tee $OUT/stagger_ab.txt
timeout 900 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-profile-pass" "FLUID_CHAIN_PERSIST=0" "FLUID_CHAIN_STACK=1" "FLUID_CHAIN_STACK=1 FLUID_CHAIN_STAGGER_US=16" "FLUID_CHAIN_STACK=2 FLUID_CHAIN_STAGGER_US=30" 2>&1 | tee -a $OUT/stagger_ab.txt
