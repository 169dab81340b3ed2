#!/bin/bash
# round 3, visit 7: the headline lines again with the parity check behind every timed section
OUT=$PWD/gpurun_out/r03final; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
FLUID_BENCH_KEEP_PMC="$OUT" timeout 1200 python bench.py >"$OUT/bench.json" 2>"$OUT/bench.err"; cut -c1-300 $OUT/bench.json
timeout 900 python bench.py --steps 20 --warmup 5 >"$OUT/bench_driver_flags.json" 2>>"$OUT/bench.err"; cut -c1-300 $OUT/bench_driver_flags.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o ks -- \
    python "$GRAFT_REPO_ROOT/bench.py" --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass >"$OUT/bench_under_rocprof.json" 2>"$OUT/rocprof.err" )
KS=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats.csv" && head -6 "$OUT/kernel_stats.csv" | cut -c1-200
rm -rf "$OUT/prof"
timeout 300 python -m pytest tests/test_bench_live.py -m gpu -x -q 2>&1 | tail -3
