#!/bin/bash
# round 5, visit 13: the chained Jacobi launch once more — the three counters polled by three lanes at once, tickets off (blockIdx order) / on
OUT=$PWD/gpurun_out/r05v13; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 100 --warmup 30 --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=4" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=2" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=0" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=4 FLUID_CHAIN_TICKET=1" 2>&1 | tee $OUT/jacobi_chain_parallel_poll.txt
timeout 300 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=4" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=0" 2>&1 | tee -a $OUT/jacobi_chain_parallel_poll.txt
timeout 300 python tools/ab_env.py --rounds 1 --args "--size 8192 --steps 40 --warmup 10 --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=4" 2>&1 | tee -a $OUT/jacobi_chain_parallel_poll.txt
