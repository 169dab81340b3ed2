#!/bin/bash
# visit 44: shapes of the two-texel Jacobi tile (FLUID_TB2 = waves,rows) at 512^2 / 1024^2 / 1536^2
OUT=gpurun_out/r03v44; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
run() { env $1 timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-16s %-36s] %8.1f steps/s %.4f ms/step  jacobi launch %.1f us  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],d['roofline']['avg_launch_ms']*1e3,{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for a in "--size 1024 --steps 2000 --warmup 200" "--size 512 --steps 4000 --warmup 400" "--size 1536 --steps 1000 --warmup 100"; do
for t in 8,5 8,4 8,6 4,10 16,3 8,5; do
run FLUID_TB2=$t "$a"
done
done
