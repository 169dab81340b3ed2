#!/bin/bash
# round 6, visit 27: with the rotation in, the two level lab forms of the tile once more on the whole step (divergence loads in front of the poll; stale-row skipping)
OUT=$PWD/gpurun_out/r06v27; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python tools/ab_env.py --rounds 5 --args "--steps 100 --warmup 30 --no-profile-pass --no-parity" "FLUID_CHAIN_ROT=5" "FLUID_CHAIN_DFIRST=1" "FLUID_CHAIN_SKIP=1" 2>&1 | tee $OUT/dfirst_skip_ab.txt
