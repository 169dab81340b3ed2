#!/bin/bash
# round 4, visit 8: texels per thread of the packed-dye advection (lab knob), then the round's canonical profile set on the final build
OUT=$PWD/gpurun_out/r04v8; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 200 --warmup 50 --no-parity" "FLUID_SKIP_CURL=1" "FLUID_ADVECT_ROWS=2" "FLUID_ADVECT_ROWS=3" "FLUID_ADVECT_ROWS=6" "FLUID_ADVECT_ROWS=8" > $OUT/ab_rgb_rows_4096.txt 2>&1; cat $OUT/ab_rgb_rows_4096.txt
timeout 1500 python -m pytest tests -m gpu -q -rsx > gpurun_out/r04final2_pytest_gpu.txt 2>&1; mkdir -p gpurun_out/r04final2; mv gpurun_out/r04final2_pytest_gpu.txt gpurun_out/r04final2/pytest_gpu.txt; tail -4 gpurun_out/r04final2/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
bash tools/gpu_round.sh r04final2 quick > $OUT/gpu_round.log 2>&1; tail -30 $OUT/gpu_round.log | cut -c1-300
