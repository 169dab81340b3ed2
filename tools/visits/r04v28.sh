#!/bin/bash
# round 4, visit 28: the in-process group with default-priority comm streams again (priority only for RCCL contexts): one-GPU decomposition
# overhead of the final schedule; stripes / tiles parity
OUT=$PWD/gpurun_out/r04v28; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for r in 1 2; do
echo -n "tiles2x2 " | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 300 python tools/bench_group.py 4096 50 56 4 2 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
echo -n "stripes4 " | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 300 python tools/bench_group.py 4096 50 56 4 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
done
echo -n "stripes2 " | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 300 python tools/bench_group.py 4096 50 56 2 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
echo -n "stripes4 cover off (lab) " | tee -a $OUT/decomposition_overhead_one_gpu.txt
FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so FLUID_COVER_JACOBI=0 timeout 300 python tools/bench_group.py 4096 50 56 4 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_hip_f16.py -m gpu -q -x > $OUT/pytest_stripes.txt 2>&1; tail -3 $OUT/pytest_stripes.txt
timeout 400 python tools/overlap_vs_link.py --quick --rounds 1 --config tile > $OUT/overlap_tile.txt 2>&1; grep "ms/step" $OUT/overlap_tile.txt | cut -c1-110
