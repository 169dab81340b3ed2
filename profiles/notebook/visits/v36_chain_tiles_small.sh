#!/bin/bash
# visit 36: tile shape of k_advect_cvd on small grids (where fluid_step_n uses it by default)
OUT=gpurun_out/r03v36; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
run() { env $1 timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-24s %-36s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for a in "--size 512 --steps 4000 --warmup 400" "--size 1024 --steps 2000 --warmup 200" "--size 2048 --steps 800 --warmup 100"; do
for t in 8,8,3 4,8,3 4,8,4 8,8,4 16,8,4; do
run FLUID_CHAIN_TILE=$t "$a"
done
run FLUID_CHAIN=0 "$a"
done
