#!/bin/bash
# round 6: the driver's exact command, seven times in a row on one lease.  Usage: bash tools/visits/r05_driver_lease.sh <lease number>
L=${1:-3}; OUT=$PWD/gpurun_out/r06lease$L; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  bench.py: $(sha256sum bench.py | cut -c1-16)"
for i in 1 2 3 4 5 6 7; do timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2>$OUT/driver_cmd_$i.err; echo "run $i exit $?"; python -c "
import json; d=json.load(open('$OUT/driver_cmd_$i.json')); print(d.get('ms_per_step'), d.get('value'), d.get('roofline',{}).get('frac'), d.get('parity_in_run',{}).get('ok'), 'cpu_baseline' in d, d.get('preloaded_window',{}).get('ms_per_step'), d.get('error'))"; done
