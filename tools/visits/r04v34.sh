#!/bin/bash
# round 4, visit 34: bench.py (changed flags) and the rebuilt addon on the GPU: tests/test_bench_live.py, tests/test_node_shim.py
OUT=$PWD/gpurun_out/r04v34; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 800 python -m pytest tests/test_bench_live.py tests/test_node_shim.py -m gpu -q -x -rsx > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
