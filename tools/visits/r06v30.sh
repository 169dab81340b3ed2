#!/bin/bash
# round 6, visit 30: the tile-shape table once more where the loop is bound by its ARITHMETIC (8192^2: chain_loop_map.txt) — round 1-3 swept it at 4096^2
OUT=$PWD/gpurun_out/r06v30; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python tools/bench_jacobi.py 8192 50 2>&1 | tee $OUT/variants_8192.txt
