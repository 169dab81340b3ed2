#!/bin/bash
# On the GPU box: rebuild libfluid_hip.so with different compiler flags and run the headline bench with each build.
# Usage: bash tools/ab_build_bench.sh <tag> "<flagsA>" "<flagsB>" ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
PKG=webgl-fluid-simulation_amd
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
i=0
for FL in "$@"; do
  i=$((i+1)); D=/tmp/abb_$i; rm -rf $D; mkdir -p $D
  ( cd $PKG && /opt/rocm/bin/hipcc $CF $FL -c csrc/fluid_kernels.hip -o $D/k.o && /opt/rocm/bin/hipcc $CF -c csrc/fluid_solver.cpp -o $D/s.o \
    && /opt/rocm/bin/hipcc $CF -c csrc/fluid_stripes.cpp -o $D/t.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libfluid_hip.so $D/k.o $D/s.o $D/t.o -ldl ) 2>&1 | grep -E "error" | head -3
  FLUID_HIP_LIB=$D/libfluid_hip.so python bench.py --steps 60 --warmup 10 --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('flags=[$FL] steps/s %.1f  ms/step %.4f  passes %s' % (d['steps_per_sec'], d['ms_per_step'], {k:v for k,v in p.items() if v}))" | tee -a $OUT/ab.txt
done
