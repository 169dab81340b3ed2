#!/bin/bash
# visit 37: apron waves of the Jacobi tile skip the row updates that can no longer reach a stored row (jacobi_sweep TRAP): parity, then A/B against -DFLUID_JACOBI_TRAP=0
OUT=gpurun_out/r03v37; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_golden.py tests/test_hip_f16.py tests/test_stripes_gpu.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1; tail -3 $OUT/pytest_subset.txt
run() { env FLUID_HIP_LIB=${1:+$PWD/build_ab/$1/libfluid_hip.so} timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-7s %-36s] %8.1f steps/s %.4f ms/step  jacobi launch %.1f us  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],d['roofline']['avg_launch_ms']*1e3,{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for i in 1 2 3; do
run notrap ""
run "" ""
done
for a in "--size 1024 --steps 2000 --warmup 200" "--size 2048 --steps 800 --warmup 100" "--size 8192 --steps 60 --warmup 10"; do
run notrap "$a"
run "" "$a"
done
