#!/bin/bash
# visit 40: k_advect_both_fast<4> held to 96 VGPRs (5 waves per SIMD) against the build before (98 VGPRs: 4 waves per SIMD)
OUT=gpurun_out/r03v40; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "fused_equals_passes" > $OUT/pytest_subset.txt 2>&1; tail -2 $OUT/pytest_subset.txt
run() { env FLUID_HIP_LIB=${1:+$PWD/build_ab/$1/libfluid_hip.so} timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-8s %-36s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for i in 1 2 3; do
run advwpe1 ""
run "" ""
done
run advwpe1 "--size 8192 --steps 60 --warmup 10"
run "" "--size 8192 --steps 60 --warmup 10"
