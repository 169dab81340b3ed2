#!/bin/bash
# round 4, visit 37: the last library of the round (sha256 f1c0923541f7c18f...): smoke and the driver's command
OUT=$PWD/gpurun_out/r04v37; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2>$OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_driver_flags.json')); print(d['ms_per_step'], d['steps_per_sec'], d['value'], d['roofline']['frac'], d['step_hbm']['frac'], d.get('parity_in_run',{}).get('ok'), 'cpu_baseline' in d)"
