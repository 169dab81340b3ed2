#!/bin/bash
# visit 42: the Jacobi tile with two texels per lane (shape 20, k_jacobi_tb2): parity with the shape forced, then A/B on small grids
OUT=gpurun_out/r03v42; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
FLUID_TB_VARIANT=20 timeout 900 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_golden.py tests/test_hip_vs_oracle.py tests/test_stripes_gpu.py -m gpu -x -q > $OUT/pytest_forced20.txt 2>&1; tail -3 $OUT/pytest_forced20.txt
run() { env $1 timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-20s %-36s] %8.1f steps/s %.4f ms/step  jacobi launch %.1f us  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],d['roofline']['avg_launch_ms']*1e3,{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for a in "--size 512 --steps 4000 --warmup 400" "--size 1024 --steps 2000 --warmup 200" "--size 1536 --steps 1000 --warmup 100" "--size 2048 --steps 800 --warmup 100"; do
for i in 1 2; do
run FLUID_TB_VARIANT=8 "$a"
run FLUID_TB_VARIANT=20 "$a"
done
done
