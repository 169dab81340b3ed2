#!/bin/bash
# round 4, visit 15: the randomized call-sequence test
OUT=$PWD/gpurun_out/r04v15; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_hip_properties.py -m gpu -q -rsx -k "random_call_sequences" > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
