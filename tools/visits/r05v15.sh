#!/bin/bash
# round 5, visit 15: the chained Jacobi launch (band-cyclic, rows per band = 64 / tiles per row) at the other widths
OUT=$PWD/gpurun_out/r05v15; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
for cfg in "3072 200 50" "4096 100 30" "6144 60 20" "8192 40 10"; do set -- $cfg
  echo "== $1^2 / 50 =="
  timeout 600 python tools/ab_env.py --rounds 2 --args "--size $1 --steps $2 --warmup $3 --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee -a $OUT/jacobi_chain_sizes.txt
done
echo "== 8192^2: bands 1 and 2; 16384^2 / 200 =="
timeout 600 python tools/ab_env.py --rounds 1 --args "--size 8192 --steps 40 --warmup 10 --no-profile-pass" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=2" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=3" 2>&1 | tee -a $OUT/jacobi_chain_sizes.txt
timeout 600 python tools/ab_env.py --rounds 1 --args "--size 16384 --iters 200 --steps 10 --warmup 3 --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee -a $OUT/jacobi_chain_sizes.txt
