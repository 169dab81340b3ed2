#!/bin/bash
# round 4, visit 10: the packed dye on the dye != sim path (sim 1024 / dye 4096: VERDICT r03 item 5), the whole suite, A/B
OUT=$PWD/gpurun_out/r04v10; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -rsx -x > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 900 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_DYE_PACK=0" > $OUT/ab_dye_ne_sim_packed.txt 2>&1; cat $OUT/ab_dye_ne_sim_packed.txt
timeout 600 python tools/bench_shipping.py > $OUT/bench_shipping_defaults.json 2> $OUT/ship.err; python -c "
import json; d=json.load(open('$OUT/bench_shipping_defaults.json'))
for r in d['kernels_fast']: print(r['case'], r['latency_us_per_step_median'], r['throughput_us_per_step'], r['pass_us_per_step'])
print('bitwise fast vs general', d['bitwise_equal_fast_vs_general'])"
