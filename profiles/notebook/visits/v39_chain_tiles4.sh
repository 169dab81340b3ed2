#!/bin/bash
# visit 39: k_advect_cvd with four rows per wave on the small grids
OUT=gpurun_out/r03v39; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "knob" > $OUT/pytest_knob.txt 2>&1; tail -2 $OUT/pytest_knob.txt
run() { env $1 timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-24s %-36s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for a in "--size 512 --steps 4000 --warmup 400" "--size 1024 --steps 2000 --warmup 200" "--size 2048 --steps 800 --warmup 100"; do
for t in 4,8,3 8,4,3 4,4,3 16,4,3 4,8,3 8,4,3; do
run FLUID_CHAIN_TILE=$t "$a"
done
done
