#!/bin/bash
# round 3, GPU visit 1: parity of the folded gradient subtract + the small-grid tile shapes, first A/Bs, RCCL init probe
OUT=$PWD/gpurun_out/r03v1; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" | tee $OUT/log.txt; tail -4 $OUT/pytest_gpu.txt | tee -a $OUT/log.txt
echo "== driver flags ==" | tee -a $OUT/log.txt
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic > $OUT/bench_driver_flags.json 2>$OUT/err.txt; cut -c1-400 $OUT/bench_driver_flags.json | tee -a $OUT/log.txt
echo "== fold A/B 4096 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50" "FLUID_FOLD_GRADSUB=1" "FLUID_FOLD_GRADSUB=0" 2>&1 | tee -a $OUT/ab_fold.txt
echo "== variants 4096 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--steps 200 --warmup 50" "FLUID_TB_VARIANT=0" "FLUID_TB_VARIANT=10" "FLUID_TB_VARIANT=3" 2>&1 | tee -a $OUT/ab_variants_4096.txt
echo "== variants 1024 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 1024 --steps 3000 --warmup 300" "" "FLUID_TB_VARIANT=0" "FLUID_TB_VARIANT=8" "FLUID_TB_VARIANT=9" "FLUID_TB_VARIANT=10" "FLUID_TB_VARIANT=11" "FLUID_TB_VARIANT=8 FLUID_ADVECT_ROWS=2" "FLUID_TB_VARIANT=8 FLUID_ADVECT_ROWS=1" 2>&1 | tee -a $OUT/ab_variants_1024.txt
echo "== variants 2048 ==" | tee -a $OUT/log.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 2048 --steps 1000 --warmup 100" "" "FLUID_TB_VARIANT=0" "FLUID_TB_VARIANT=8" "FLUID_TB_VARIANT=9" "FLUID_TB_VARIANT=10" 2>&1 | tee -a $OUT/ab_variants_2048.txt
echo "== rccl init probe ==" | tee -a $OUT/log.txt
timeout 400 python tools/rccl_init_probe.py --trials 3 --timeout 40 > $OUT/rccl_init_probe.txt 2>&1; grep "^==" $OUT/rccl_init_probe.txt | tee -a $OUT/log.txt
echo "== done =="
