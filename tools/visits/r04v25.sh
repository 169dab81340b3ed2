#!/bin/bash
# round 4, visit 25: the pressure loop as two row chains on two streams (lab knob FLUID_JACOBI_CHAINS): same bits, then A/B at 4096^2
OUT=$PWD/gpurun_out/r04v25; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_hip_properties.py -m gpu -q -x -k "chain or knob" > $OUT/pytest_knobs.txt 2>&1; tail -3 $OUT/pytest_knobs.txt
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAINS=0.5" "FLUID_JACOBI_CHAINS=0.4" "FLUID_JACOBI_CHAINS=0.6" > $OUT/ab_jacobi_chains.txt 2>&1; cut -c1-120 $OUT/ab_jacobi_chains.txt
