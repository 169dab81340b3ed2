#!/bin/bash
OUT=gpurun_out/r03v29; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_golden.py tests/test_hip_vs_oracle.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1; tail -5 $OUT/pytest_subset.txt
for i in 1 2 3; do
for lib in "" build_ab/fma/libfluid_hip.so; do
  FLUID_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$lib] %.1f steps/s %.4f ms/step jacobi launch %.1f us'%(d['steps_per_sec'],d['ms_per_step'],d['roofline']['avg_launch_ms']*1e3))" | tee -a $OUT/ab.txt
done; done
