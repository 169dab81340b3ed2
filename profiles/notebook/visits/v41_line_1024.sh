#!/bin/bash
# visit 41: the full bench line (in-run PMC traffic, SQ counters, parity) and the rocprofv3 kernel stats at BASELINE configs[1]'s grid, 1024^2 / 50
OUT=$PWD/gpurun_out/r03v41; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python bench.py --size 1024 --steps 2000 --warmup 200 --cpu-budget 0 > $OUT/bench_1024_50.json 2> $OUT/err.txt
tail -c 600 $OUT/bench_1024_50.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --size 1024 --steps 2000 --warmup 200 --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass > $OUT/bench_1024_under_rocprof.json 2>>$OUT/err.txt )
KS=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$KS" ] && cp $KS $OUT/kernel_stats_1024.csv && head -8 $OUT/kernel_stats_1024.csv | cut -c1-160
rm -rf $OUT/prof
