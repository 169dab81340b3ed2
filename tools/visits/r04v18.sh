#!/bin/bash
# round 4, visit 18: the first Jacobi launch as cover of the step's first exchange, on / off, three rounds each, alternating
OUT=$PWD/gpurun_out/r04v18; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for r in 1 2 3; do
  timeout 400 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_on_$r.txt 2>&1
  FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so FLUID_COVER_JACOBI=0 timeout 400 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_off_$r.txt 2>&1
done
for f in $OUT/overlap_on_*.txt $OUT/overlap_off_*.txt; do echo "== $f"; grep "ms/step" $f | cut -c1-70; done
