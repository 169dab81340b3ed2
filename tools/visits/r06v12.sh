#!/bin/bash
# round 6, visit 12: WHY the chained launch loses where the pressure set no longer fits the Infinity Cache (8192^2: 768 MB) — its timing probes there
# (results of the probe rows are not valid, only their times): 2 = tiles count themselves without draining their write-through stores, 3 = plain
# loads / stores instead of sc1, 4 = nobody waits, 5 = no arithmetic
OUT=$PWD/gpurun_out/r06v12; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1200 python tools/ab_env.py --rounds 2 --args "--size 8192 --steps 40 --warmup 10 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=2" "FLUID_JACOBI_CHAIN=3" "FLUID_JACOBI_CHAIN=4" "FLUID_JACOBI_CHAIN=5" 2>&1 | tee $OUT/probes_8192.txt
