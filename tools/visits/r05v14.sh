#!/bin/bash
# round 5, visit 14: the chained Jacobi launch, band-cyclic order with 1 / 2 / 3 rows per band, and what each order does with nobody waiting
OUT=$PWD/gpurun_out/r05v14; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 100 --warmup 30 --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=2" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=3" 2>&1 | tee $OUT/jacobi_chain_bands123.txt
timeout 300 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=1" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=2" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=3" 2>&1 | tee -a $OUT/jacobi_chain_bands123.txt
