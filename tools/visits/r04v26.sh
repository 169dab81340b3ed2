#!/bin/bash
# round 4, visit 26: the final tree — the whole GPU suite, smoke, the driver's flags, the shipping shapes, the one-GPU decomposition overhead
OUT=$PWD/gpurun_out/r04v26; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>$OUT/bench.err; cut -c1-400 $OUT/bench_driver_flags.json
timeout 600 python tools/bench_shipping.py > $OUT/bench_shipping_defaults.json 2> $OUT/ship.err; python -c "
import json; d=json.load(open('$OUT/bench_shipping_defaults.json'))
for r in d['kernels_fast']: print(r['case'], r['latency_us_per_step_median'], r['throughput_us_per_step'], r['pass_us_per_step'])
print('bitwise fast vs general', d['bitwise_equal_fast_vs_general'])"
for r in 1 2; do
echo -n "tiles2x2 " | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 300 python tools/bench_group.py 4096 50 56 4 2 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
echo -n "stripes4 " | tee -a $OUT/decomposition_overhead_one_gpu.txt
timeout 300 python tools/bench_group.py 4096 50 56 4 1 2>>$OUT/err.txt | tail -1 | tee -a $OUT/decomposition_overhead_one_gpu.txt
done
