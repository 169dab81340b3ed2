#!/bin/bash
# round 3, GPU visit 2: the new bench line, small-grid shapes (deeper aprons), shipping defaults, kernel trace, then the whole GPU suite
OUT=$PWD/gpurun_out/r03v2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== bench default ==" | tee $OUT/log.txt
FLUID_BENCH_KEEP_PMC="$OUT" timeout 900 python bench.py > $OUT/bench_4096_50.json 2> $OUT/bench.err; echo "exit $?" | tee -a $OUT/log.txt; cut -c1-1500 $OUT/bench_4096_50.json | tee -a $OUT/log.txt
echo "== driver flags ==" | tee -a $OUT/log.txt
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic > $OUT/bench_4096_50_steps20_warmup5.json 2>>$OUT/bench.err; cut -c1-330 $OUT/bench_4096_50_steps20_warmup5.json | tee -a $OUT/log.txt
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-parity > $OUT/bench_4096_50_steps20_warmup5_noparity.json 2>>$OUT/bench.err; cut -c1-330 $OUT/bench_4096_50_steps20_warmup5_noparity.json | tee -a $OUT/log.txt
echo "== shapes 1024 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 1024 --steps 3000 --warmup 300 --no-parity" "" "FLUID_FOLD_GRADSUB=0" "FLUID_TB_VARIANT=11" "FLUID_TB_VARIANT=12" "FLUID_TB_VARIANT=13" "FLUID_TB_VARIANT=14" "FLUID_TB_VARIANT=15" "FLUID_TB_VARIANT=14 FLUID_FOLD_GRADSUB=0" 2>&1 | tee $OUT/ab_shapes_1024.txt
echo "== shapes 2048 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 2048 --steps 1000 --warmup 100 --no-parity" "" "FLUID_FOLD_GRADSUB=0" "FLUID_TB_VARIANT=12" "FLUID_TB_VARIANT=13" "FLUID_TB_VARIANT=14" "FLUID_TB_VARIANT=15" 2>&1 | tee $OUT/ab_shapes_2048.txt
echo "== 1024 line ==" | tee -a $OUT/log.txt
timeout 600 python bench.py --size 1024 --steps 3000 --warmup 300 --cpu-budget 0 > $OUT/bench_1024_50.json 2>>$OUT/bench.err; cut -c1-1200 $OUT/bench_1024_50.json | tee -a $OUT/log.txt
echo "== shipping defaults ==" | tee -a $OUT/log.txt
timeout 600 python tools/bench_shipping.py > $OUT/bench_shipping_defaults.json 2>>$OUT/bench.err; cat $OUT/bench_shipping_defaults.json | tee -a $OUT/log.txt
echo "== rocprofv3 kernel trace ==" | tee -a $OUT/log.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o ks -- \
    python "$GRAFT_REPO_ROOT/bench.py" --cpu-budget 0 --no-traffic --no-steady --no-parity >"$OUT/bench_under_rocprof.json" 2>"$OUT/rocprof.err" )
KS=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats_fused_4096_50.csv" && head -12 "$OUT/kernel_stats_fused_4096_50.csv" | cut -c1-200 | tee -a $OUT/log.txt
rm -rf "$OUT/prof"
echo "== pytest ==" | tee -a $OUT/log.txt
timeout 1200 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" | tee -a $OUT/log.txt; tail -25 $OUT/pytest_gpu.txt | cut -c1-300 | tee -a $OUT/log.txt
echo "== done =="
