#!/bin/bash
# round 2, visit 10: bisect the node tile-rank hang inside the full suite
set -u
OUT=$PWD/gpurun_out/r02_v10; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== node tests after test_long_horizon ==" | tee $OUT/log.txt
NCCL_DEBUG=INFO timeout 900 python -m pytest tests/test_long_horizon.py tests/test_node_shim.py -m gpu -x -q 2>&1 | tail -40 | tee -a $OUT/log.txt
echo "== node tests after test_hip_f16 ==" | tee -a $OUT/log.txt
NCCL_DEBUG=INFO timeout 900 python -m pytest tests/test_hip_f16.py tests/test_node_shim.py -m gpu -x -q 2>&1 | tail -40 | tee -a $OUT/log.txt
echo "== node tests after test_baseline_sizes + test_display ==" | tee -a $OUT/log.txt
NCCL_DEBUG=INFO timeout 900 python -m pytest tests/test_baseline_sizes.py tests/test_display.py tests/test_node_shim.py -m gpu -x -q 2>&1 | tail -40 | tee -a $OUT/log.txt
ls /dev/shm | head -20 | tee -a $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
