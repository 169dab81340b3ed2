#!/bin/bash
# On the GPU box: rebuild libfluid_hip.so with different compiler flags and run the headline bench with each build.
# Usage: bash tools/ab_build_bench.sh <tag> "<flagsA>" "<flagsB>" ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
PKG=webgl-fluid-simulation_amd
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
i=0
for FL in "$@"; do
  i=$((i+1))
  D=/tmp/abb_$i; rm -rf $D; mkdir -p $D; cp -r $PKG $D/pkg; cp -r include $D/include
  make -C $D/pkg clean >/dev/null; make -C $D/pkg -j4 EXTRA="$FL" 2>&1 | grep -E "error" | head -3
  FLUID_HIP_LIB=$D/pkg/libfluid_hip.so python bench.py --steps 60 --warmup 10 --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('flags=[$FL] steps/s %.1f  ms/step %.4f  passes %s' % (d['steps_per_sec'], d['ms_per_step'], {k:v for k,v in p.items() if v}))" | tee -a $OUT/ab.txt
done
