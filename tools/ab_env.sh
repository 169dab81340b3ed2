#!/bin/bash
# On the GPU box: sweep one environment knob of libfluid_hip.so over the Jacobi micro-benchmark.
# Usage: bash tools/ab_env.sh <tag> <ENVVAR> "<values>" "<variants>" [iters]
TAG=$1; VAR=$2; VALS=$3; VARS=$4; IT=${5:-50}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for V in $VALS; do
  echo "=== $VAR=$V iters $IT ===" | tee -a $OUT/env.txt
  env $VAR=$V TB_VARIANTS="$VARS" python tools/bench_jacobi.py 4096 $IT 2>&1 | tee -a $OUT/env.txt
done
