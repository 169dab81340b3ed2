#!/bin/bash
# round 4, visit 36: the cut geometry moved into csrc/fluid_cut.h — stripes / tiles / fp16 groups again
OUT=$PWD/gpurun_out/r04v36; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_stripes_gpu.py tests/test_hip_f16.py tests/test_baseline_sizes.py -m gpu -q -x > $OUT/pytest_stripes.txt 2>&1; tail -4 $OUT/pytest_stripes.txt
