#!/bin/bash
# visit 35: the whole GPU suite with k_advect_cvd in fluid_step_n, then the chain on / off at 1024^2, 2048^2, 4096^2 and under the driver's flags
OUT=gpurun_out/r03v35; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
run() { env $1 timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-14s %-36s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for i in 1 2; do
for a in "--size 1024 --steps 2000 --warmup 200" "--size 2048 --steps 800 --warmup 100" "" "--steps 20 --warmup 5"; do
run FLUID_CHAIN=0 "$a"
run FLUID_CHAIN=1 "$a"
done; done
