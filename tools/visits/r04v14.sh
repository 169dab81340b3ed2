#!/bin/bash
# round 4, visit 14: dye grid != sim grid chained (velocity advection + next curl / vorticity / divergence in one launch): the whole suite,
# then the reference's shipping configuration and its 8x sibling (tools/bench_shipping.py) and the per-frame path
OUT=$PWD/gpurun_out/r04v14; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -rsx -x > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 600 python tools/bench_shipping.py > $OUT/bench_shipping_defaults.json 2> $OUT/ship.err; python -c "
import json; d=json.load(open('$OUT/bench_shipping_defaults.json'))
for r in d['kernels_fast']: print(r['case'], 'latency', r['latency_us_per_step_median'], 'throughput', r['throughput_us_per_step'], r['pass_us_per_step'])
print('bitwise fast vs general', d['bitwise_equal_fast_vs_general'])"
timeout 300 python tools/ab_passes.py --sim 128 --dye 1024 --iters 20 --steps 4000 --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_RUN_AHEAD=0" > $OUT/ab_shipping_chain.txt 2>&1; cat $OUT/ab_shipping_chain.txt
timeout 300 python tools/ab_passes.py --sim 256 --dye 2048 --iters 20 --steps 2000 --rounds 2 "FLUID_SKIP_CURL=1" "FLUID_RUN_AHEAD=0" >> $OUT/ab_shipping_chain.txt 2>&1; tail -5 $OUT/ab_shipping_chain.txt
