#!/bin/bash
# round 6, visit 29: the GPU suite and the smoke on the final library once more on another box (flakiness), then the driver's command three times
OUT=$PWD/gpurun_out/r06v29; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2>$OUT/driver_cmd_$i.err; echo "run $i exit $?"; python -c "
import json; d=json.load(open('$OUT/driver_cmd_$i.json')); print(d.get('ms_per_step'), d.get('value'), d.get('roofline',{}).get('frac'), d.get('parity_in_run',{}).get('ok'), 'cpu_baseline' in d, d.get('error'))"; done
