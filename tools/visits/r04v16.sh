#!/bin/bash
# round 4, visit 16: the whole suite on the final tree (with the randomized sequences)
OUT=$PWD/gpurun_out/r04v16; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-budget 0 2>/dev/null | cut -c1-600
