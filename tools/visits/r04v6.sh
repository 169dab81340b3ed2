#!/bin/bash
# round 4, visit 6: the comm stream at the highest stream priority: stripe / tile parity, then the one-rank loopback probe again
OUT=$PWD/gpurun_out/r04v6; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py -m gpu -q -rsx > $OUT/pytest_stripes.txt 2>&1; tail -3 $OUT/pytest_stripes.txt
timeout 1500 python tools/overlap_vs_link.py --rounds 1 > $OUT/overlap_vs_link_latency.txt 2>&1; cat $OUT/overlap_vs_link_latency.txt
