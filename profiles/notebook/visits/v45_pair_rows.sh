#!/bin/bash
# visit 45: the two-texel Jacobi tile with 4 / 6 rows per wave forced through the parity tests, then the sizes whose default changed
OUT=gpurun_out/r03v45; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_hip_properties.py -m gpu -x -q -k knob > $OUT/pytest_knob.txt 2>&1; tail -2 $OUT/pytest_knob.txt
for t in 8,4 8,6; do
FLUID_TB2=$t FLUID_TB_VARIANT=20 timeout 600 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_golden.py tests/test_stripes_gpu.py -m gpu -x -q -k "not knob and not bench_size" > $OUT/pytest_forced_${t/,/x}.txt 2>&1; tail -1 $OUT/pytest_forced_${t/,/x}.txt
done
run() { timeout 300 python bench.py $1 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-36s] %8.1f steps/s %.4f ms/step  jacobi launch %.1f us'%('$1',d['steps_per_sec'],d['ms_per_step'],d['roofline']['avg_launch_ms']*1e3))" | tee -a $OUT/ab.txt; }
run "--size 512 --steps 4000 --warmup 400"
run "--size 1024 --steps 2000 --warmup 200"
run "--size 1536 --steps 1000 --warmup 100"
