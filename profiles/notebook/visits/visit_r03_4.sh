#!/bin/bash
# round 3, GPU visit 4: K6 on the fly inside the advection (k_project_advect): parity, A/B; the step time against the step index
OUT=$PWD/gpurun_out/r03v4; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== parity ==" | tee $OUT/log.txt
timeout 900 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_oracle.py tests/test_hip_vs_golden.py tests/test_big_passes_4096.py tests/test_long_horizon.py -m gpu -x -q > $OUT/pytest_project.txt 2>&1; echo "exit $?" | tee -a $OUT/log.txt; tail -5 $OUT/pytest_project.txt | cut -c1-300 | tee -a $OUT/log.txt
echo "== 4096: K6 on the fly (rows 4 / 3 / 2) against K6 as a pass ==" | tee -a $OUT/log.txt
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_PROJECT_ADVECT=0" "FLUID_PROJECT_ROWS=4" "FLUID_PROJECT_ROWS=3" "FLUID_PROJECT_ROWS=2" 2>&1 | tee $OUT/ab_project_4096.txt
echo "== 1024 / 2048 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 1024 --steps 3000 --warmup 300 --no-parity" "FLUID_PROJECT_ADVECT=0" "FLUID_PROJECT_ROWS=4" "FLUID_PROJECT_ROWS=2" "FLUID_PROJECT_ROWS=1" 2>&1 | tee $OUT/ab_project_1024.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 2048 --steps 1000 --warmup 100 --no-parity" "FLUID_PROJECT_ADVECT=0" "FLUID_PROJECT_ROWS=4" "FLUID_PROJECT_ROWS=2" 2>&1 | tee $OUT/ab_project_2048.txt
timeout 600 python tools/ab_env.py --rounds 1 --args "--size 8192 --steps 60 --warmup 10 --no-parity" "FLUID_PROJECT_ADVECT=0" "FLUID_PROJECT_ROWS=4" "FLUID_PROJECT_ROWS=3" 2>&1 | tee $OUT/ab_project_8192.txt
echo "== step timeline ==" | tee -a $OUT/log.txt
timeout 300 python tools/step_timeline.py 4096 50 2>&1 | grep "^steps" | tee $OUT/step_timeline_4096.txt
echo "== done =="
