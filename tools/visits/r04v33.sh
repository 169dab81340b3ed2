#!/bin/bash
# round 4, visit 33: fluid_set_link_model — the stripe / tile tests incl. the new link-model cases
OUT=$PWD/gpurun_out/r04v33; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_stripes_gpu.py tests/test_abi.py -m gpu -q -x > $OUT/pytest_stripes.txt 2>&1; tail -4 $OUT/pytest_stripes.txt
