#!/bin/bash
# visit 32: k_advect_cvd v2 (packed pair arithmetic, interior tiles without border selects): parity subset, A/B against FLUID_CHAIN=0, tile shapes
OUT=gpurun_out/r03v33; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_properties.py -m gpu -x -q > $OUT/pytest_subset.txt 2>&1; tail -3 $OUT/pytest_subset.txt
run() { env $1 timeout 300 python bench.py --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-28s] %8.1f steps/s %.4f ms/step  passes(us) %s'%('$1',d['steps_per_sec'],d['ms_per_step'],{k[:-3]:round(v*1e3,1) for k,v in d['pass_ms_per_step'].items() if v}))" | tee -a $OUT/ab.txt; }
for i in 1 2; do
run FLUID_CHAIN=0
run FLUID_CHAIN=1
run FLUID_CHAIN_TILE=8,8,3
run FLUID_CHAIN_TILE=4,8,4
run FLUID_CHAIN_TILE=16,8,4
done
