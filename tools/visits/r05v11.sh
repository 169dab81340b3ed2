#!/bin/bash
# round 5, visit 11: where the chained Jacobi launch loses its time — its three timing probes (results invalid, hence --no-parity) against the
# chain proper and the shipped five launches, and the plain (unmixed) five launches it should be compared with (FLUID_TB_TAIL=0,0,7: no small tiles).
OUT=$PWD/gpurun_out/r05v11; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1200 python tools/ab_env.py --rounds 2 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_TB_TAIL=0,0,7" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=2" "FLUID_JACOBI_CHAIN=3" "FLUID_JACOBI_CHAIN=4" 2>&1 | tee $OUT/jacobi_chain_probes.txt
