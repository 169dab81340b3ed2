#!/bin/bash
# A/B on the GPU box: rebuild libfluid_hip.so with different compiler flags and time the temporally blocked
# Jacobi kernel variants (tools/bench_jacobi.py).  Usage: bash tools/ab_jacobi.sh <tag> "<variants>" "<flagsA>" "<flagsB>" ...
TAG=$1; VARS=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
PKG=webgl-fluid-simulation_amd
i=0
for FL in "$@"; do
  i=$((i+1))
  D=/tmp/ab_$i; rm -rf $D; mkdir -p $D; cp -r $PKG $D/pkg; cp -r include $D/include
  make -C $D/pkg clean >/dev/null; make -C $D/pkg -j4 EXTRA="$FL" 2>&1 | grep -E "error" | head -3
  echo "=== build $i flags=[$FL] ===" | tee -a $OUT/ab.txt
  FLUID_HIP_LIB=/tmp/ab_$i/pkg/libfluid_hip.so TB_VARIANTS="$VARS" python tools/bench_jacobi.py 4096 50 2>&1 | tee -a $OUT/ab.txt
done
