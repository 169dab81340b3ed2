#!/bin/bash
# visit 34: k_advect_cvd — all eight rows of the velocity advection in flight (in-tree build), and the dye gathers pipelined under the stencil stages (variant builds)
OUT=gpurun_out/r03v34; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "step_n or fused_equals_passes_bitwise or tiny" > $OUT/pytest_subset.txt 2>&1; tail -2 $OUT/pytest_subset.txt
FLUID_HIP_LIB=$PWD/build_ab/chp3/libfluid_hip.so FLUID_CHAIN_TILE=4,8,4 timeout 600 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k "step_n or fused_equals_passes_bitwise or tiny" > $OUT/pytest_subset_chp3.txt 2>&1; tail -2 $OUT/pytest_subset_chp3.txt
run() { env $2 FLUID_HIP_LIB=${1:+$PWD/build_ab/$1/libfluid_hip.so} timeout 300 python bench.py --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-5s %-24s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1','$2',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for i in 1 2; do
run "" FLUID_CHAIN=0
run "" FLUID_CHAIN=1
run "" FLUID_CHAIN_TILE=4,8,4
run chp3 FLUID_CHAIN_TILE=4,8,4
run chp3 FLUID_CHAIN_TILE=8,8,4
run chp2 FLUID_CHAIN=1
run chp2 FLUID_CHAIN_TILE=4,8,4
done
