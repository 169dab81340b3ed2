#!/bin/bash
# round 2, visit 7: stripe / tile tests with the pack kernel, decomposition overhead on one GPU (tiles vs stripes, halo 4 vs 56)
set -u
OUT=$PWD/gpurun_out/r02_v7; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_hip_f16.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/log.txt
echo "== four contexts of 4096^2 on one GPU: 2 x 2 tiles (8192^2) vs four stripes (4096 x 16384), halo 56 ==" | tee -a $OUT/log.txt
python tools/bench_group.py 4096 50 56 4 2 | tee -a $OUT/log.txt
python tools/bench_group.py 4096 50 56 4 1 | tee -a $OUT/log.txt
echo "== two stripes of 4096^2: halo 56 (2 exchanges per step) vs halo 4 (the literal one-exchange-per-iteration decomposition) ==" | tee -a $OUT/log.txt
python tools/bench_group.py 4096 50 56 2 1 | tee -a $OUT/log.txt
python tools/bench_group.py 4096 50 32 2 1 | tee -a $OUT/log.txt
python tools/bench_group.py 4096 50 4 2 1 | tee -a $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
