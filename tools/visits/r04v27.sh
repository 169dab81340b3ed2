#!/bin/bash
# round 4, visit 27: when did the one-GPU in-process group get slow?  tools/bench_group.py with the library of three earlier commits and HEAD
OUT=$PWD/gpurun_out/r04v27; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for sha in 4e43a7b 6c812d1 3ef61ba HEAD; do
  if [ $sha = HEAD ]; then L=$PWD/webgl-fluid-simulation_amd/libfluid_hip.so; else L=$PWD/build_ab/bis_$sha/webgl-fluid-simulation_amd/libfluid_hip.so; fi
  echo -n "$sha stripes4 " | tee -a $OUT/bisect.txt
  FLUID_HIP_LIB=$L timeout 300 python tools/bench_group.py 4096 50 56 4 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/bisect.txt
done
echo -n "HEAD stripes2 " | tee -a $OUT/bisect.txt
timeout 300 python tools/bench_group.py 4096 50 56 2 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/bisect.txt
