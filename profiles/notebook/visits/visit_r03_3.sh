#!/bin/bash
# round 3, GPU visit 3: driver flags with the parity check in front of the warm-up, dye != sim rows per thread, clocks at small grids,
# rect-list strips of 2-D tiles (A/B + the tile tests)
OUT=$PWD/gpurun_out/r03v3; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== driver flags, interleaved: parity in front of the warm-up / --no-parity ==" | tee $OUT/log.txt
for k in 1 2 3; do
  for a in "" "--no-parity"; do
    timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-profile-pass $a 2>>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[%-12s] ms/step %.4f  steady %.4f  parity %s' % ('$a', d['ms_per_step'], d.get('steady_ms_per_step', 0), (d.get('parity_in_run') or {}).get('seconds')))" | tee -a $OUT/driver_flags_preroll.txt
  done
done
echo "== tile tests ==" | tee -a $OUT/log.txt
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py -m gpu -x -q > $OUT/pytest_tiles.txt 2>&1; echo "exit $?" | tee -a $OUT/log.txt; tail -3 $OUT/pytest_tiles.txt | tee -a $OUT/log.txt
echo "== 2x2 tiles of 8192^2 on one GPU: strips as one launch (1) / one launch each (0) ==" | tee -a $OUT/log.txt
for k in 1 2; do
  for v in 1 0; do
    echo -n "FLUID_STRIP_RECTS=$v " | tee -a $OUT/decomposition_overhead_one_gpu.txt
    FLUID_STRIP_RECTS=$v timeout 300 python tools/bench_group.py 4096 50 56 4 2 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
  done
done
echo -n "stripes4 " | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 300 python tools/bench_group.py 4096 50 56 4 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
echo "== dye != sim: rows per thread of the split fast kernels ==" | tee -a $OUT/log.txt
for r in 4 2 1; do
  echo "FLUID_ADVECT_SPLIT_ROWS=$r" | tee -a $OUT/shipping_rows.txt
  FLUID_ADVECT_SPLIT_ROWS=$r timeout 300 python tools/bench_shipping.py 2>>$OUT/err.txt | python -c "
import json,sys
d=json.load(sys.stdin)
for c in d['kernels_fast']: print('   ', c['case'], 'latency', c['latency_us_per_step_median'], 'throughput', c['throughput_us_per_step'], c['pass_us_per_step'])
c=d['kernels_general_FLUID_ADVECT_FAST_0'][1]; print('    general', c['case'], 'throughput', c['throughput_us_per_step'], c['pass_us_per_step'])" | tee -a $OUT/shipping_rows.txt
done
echo "== clocks ==" | tee -a $OUT/log.txt
timeout 200 python tools/clock_probe.py 1024 2048 4096 2>&1 | tee $OUT/clock_probe.txt
echo "== done =="
