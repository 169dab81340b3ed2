#!/bin/bash
# round 2, visit 13: what would a launch cost if part of the tile's divergence were already on chip (prefetched under the previous tile's arithmetic)?
set -u
OUT=$PWD/gpurun_out/r02_v13; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for lib in "" build_ab/skipd48 build_ab/skipd80; do for it in 1 10; do
if [ -z "$lib" ]; then TB_VARIANTS=0 python tools/bench_jacobi.py 4096 $it | sed "s/^/all-loads /" | tee -a $OUT/log.txt
else FLUID_HIP_LIB=$PWD/$lib/libfluid_hip.so TB_VARIANTS=0 python tools/bench_jacobi.py 4096 $it | sed "s|^|$lib |" | tee -a $OUT/log.txt; fi
done; done
echo "== done ==" | tee -a $OUT/log.txt
