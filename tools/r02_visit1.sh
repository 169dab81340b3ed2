#!/bin/bash
# round 2, visit 1: ceilings of this box, the live reference on the GPU box, the no-tail Jacobi experiment
set -u
OUT=$PWD/gpurun_out/r02_v1; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== nproc / cpu ==" | tee $OUT/log.txt
nproc | tee -a $OUT/log.txt; grep -m1 "model name" /proc/cpuinfo | tee -a $OUT/log.txt
echo "== stream_bw ==" | tee -a $OUT/log.txt
timeout 300 tools/micro/stream_bw 256 20 2>&1 | tee $OUT/stream_bw.txt | tail -30 | tee -a $OUT/log.txt
echo "== live reference on this box ==" | tee -a $OUT/log.txt
timeout 600 python oracle/live/time_reference.py --size 4096 --iters 50 --warm 3 --timed 5 --json > $OUT/reference_timing_gpu_box.json 2> $OUT/reference_err.txt
echo "exit $?" | tee -a $OUT/log.txt; cat $OUT/reference_timing_gpu_box.json | cut -c1-600 | tee -a $OUT/log.txt; tail -3 $OUT/reference_err.txt | tee -a $OUT/log.txt
echo "== bench headline (with cpu_baseline) ==" | tee -a $OUT/log.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?" | tee -a $OUT/log.txt
cat $OUT/bench.json | tee -a $OUT/log.txt; tail -3 $OUT/bench.err
echo "== jacobi 10 iterations, one launch: normal vs 5x tiles in one launch ==" | tee -a $OUT/log.txt
for it in 1 10; do
TB_VARIANTS="0" python tools/bench_jacobi.py 4096 $it | tee -a $OUT/log.txt
FLUID_HIP_LIB=$PWD/build_ab/rep5/libfluid_hip.so TB_VARIANTS="0" python tools/bench_jacobi.py 4096 $it | sed 's/^/rep5 /' | tee -a $OUT/log.txt
done
TB_VARIANTS="4" python tools/bench_jacobi.py 4096 13 | tee -a $OUT/log.txt
FLUID_HIP_LIB=$PWD/build_ab/rep4/libfluid_hip.so TB_VARIANTS="4" python tools/bench_jacobi.py 4096 13 | sed 's/^/rep4 /' | tee -a $OUT/log.txt
TB_VARIANTS="5" python tools/bench_jacobi.py 4096 17 | tee -a $OUT/log.txt
FLUID_HIP_LIB=$PWD/build_ab/rep4/libfluid_hip.so TB_VARIANTS="5" python tools/bench_jacobi.py 4096 17 | sed 's/^/rep4 /' | tee -a $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
