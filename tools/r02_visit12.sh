#!/bin/bash
# round 2, visit 12: ONE workgroup per CU (dynamic LDS pad) — how fast does a workgroup iterate when it has the CU to itself?
set -u
OUT=$PWD/gpurun_out/r02_v12; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for pad in 0 98304; do for it in 1 4 7 10; do
FLUID_TB_LDS_PAD=$pad TB_VARIANTS=0 python tools/bench_jacobi.py 4096 $it | sed "s/^/pad$pad /" | tee -a $OUT/log.txt
done; done
echo "== done ==" | tee -a $OUT/log.txt
