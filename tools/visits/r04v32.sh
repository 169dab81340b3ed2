#!/bin/bash
# round 4, visit 32: the final tree — the whole GPU suite, smoke, the driver's flags (3 x), the default bench line
OUT=$PWD/gpurun_out/r04v32; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1800 python -m pytest tests -m gpu -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
for r in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags_$r.json 2>>$OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_driver_flags_$r.json')); print(d['ms_per_step'], d['steps_per_sec'], d['value'], d['roofline']['frac'], d['roofline'].get('frac_compulsory'), d['step_hbm']['frac'], d.get('timed_window_regime',{}).get('ms_per_timed_step',[])[:6])"; done
