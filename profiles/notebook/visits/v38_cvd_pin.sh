#!/bin/bash
# visit 38: k_curl_vort_div with the vorticity result pinned where it is computed (96 VGPRs, no scratch) against the build before (128 VGPRs, 8 B scratch)
OUT=gpurun_out/r03v38; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_golden.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1; tail -2 $OUT/pytest_subset.txt
run() { env FLUID_HIP_LIB=${1:+$PWD/build_ab/$1/libfluid_hip.so} $3 timeout 300 python bench.py $2 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-6s %-14s %-36s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1','$3','$2',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for i in 1 2 3; do
run nopin ""
run "" ""
done
run nopin "" FLUID_CVD_TAIL=0,0
run "" "" FLUID_CVD_TAIL=0,0
for a in "--size 8192 --steps 60 --warmup 10" "--size 1024 --steps 2000 --warmup 200 --schedule fused" ; do
run nopin "$a" FLUID_CHAIN=0
run "" "$a" FLUID_CHAIN=0
done
