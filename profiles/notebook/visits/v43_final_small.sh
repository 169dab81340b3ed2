#!/bin/bash
# visit 43: the GPU suite with the pair tile as the small-grid default, then the lines that change with it (other sizes, shipping defaults, the 1024^2 line + kernel stats)
OUT=$PWD/gpurun_out/r03v43; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
bash tools/other_sizes.sh r03v43 > $OUT/other.log 2>&1; cat $OUT/bench_other_sizes.txt
timeout 600 python tools/bench_shipping.py > $OUT/bench_shipping_defaults.json 2>>$OUT/err.txt; grep -m3 "throughput_us_per_step\|latency_us_per_step_median" $OUT/bench_shipping_defaults.json
timeout 600 python bench.py --size 1024 --steps 2000 --warmup 200 --cpu-budget 0 > $OUT/bench_1024_50.json 2>> $OUT/err.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --size 1024 --steps 2000 --warmup 200 --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass > $OUT/bench_1024_under_rocprof.json 2>>$OUT/err.txt )
KS=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$KS" ] && cp $KS $OUT/kernel_stats_1024.csv; rm -rf $OUT/prof
