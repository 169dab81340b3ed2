#!/bin/bash
# A/B on the GPU box: rebuild libfluid_hip.so with different compiler flags and time the temporally blocked
# Jacobi kernel variants (tools/bench_jacobi.py).  Usage: bash tools/ab_jacobi.sh <tag> "<variants>" "<flagsA>" "<flagsB>" ...
TAG=$1; VARS=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
PKG=webgl-fluid-simulation_amd
i=0
for FL in "$@"; do
  i=$((i+1))
  rm -rf /tmp/ab_$i && mkdir -p /tmp/ab_$i
  ( cd $PKG && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $FL -c csrc/fluid_kernels.hip -o /tmp/ab_$i/k.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -c csrc/fluid_solver.cpp -o /tmp/ab_$i/s.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -c csrc/fluid_stripes.cpp -o /tmp/ab_$i/t.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/ab_$i/libfluid_hip.so /tmp/ab_$i/k.o /tmp/ab_$i/s.o /tmp/ab_$i/t.o -ldl ) 2>&1 | tail -3
  echo "=== build $i flags=[$FL] ===" | tee -a $OUT/ab.txt
  FLUID_HIP_LIB=/tmp/ab_$i/libfluid_hip.so TB_VARIANTS="$VARS" python tools/bench_jacobi.py 4096 50 2>&1 | tee -a $OUT/ab.txt
done
