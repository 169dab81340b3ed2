#!/bin/bash
# round 4, visit 13: the stripes test file with the loopback-probe test, the live bench test, the node tests
OUT=$PWD/gpurun_out/r04v13; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_stripes_gpu.py tests/test_bench_live.py tests/test_node_shim.py -m gpu -q -rsx > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
