#!/bin/bash
OUT=$PWD/gpurun_out/r03v20; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for N in 4096 8192 3072; do
  ST=200; [ $N = 8192 ] && ST=60; [ $N = 3072 ] && ST=300
  echo "== $N ==" | tee -a $OUT/ab_mix_tiles.txt
  timeout 900 python tools/ab_env.py --rounds 2 --args "--size $N --steps $ST --warmup 40 --no-parity" "" "FLUID_TB_TAIL_TILES=256,512,5" "FLUID_TB_TAIL_TILES=192,384,5" "FLUID_TB_TAIL_TILES=384,640,5" "FLUID_TB_TAIL_TILES=192,384,7" "FLUID_TB_TAIL_TILES=256,512,6" 2>&1 | tee -a $OUT/ab_mix_tiles.txt
done
echo "== 16384 / 200 ==" | tee -a $OUT/ab_mix_tiles.txt
timeout 900 python tools/ab_env.py --rounds 2 --args "--size 16384 --iters 200 --steps 12 --warmup 3 --no-parity" "" "FLUID_TB_TAIL=0,0,7" "FLUID_TB_TAIL_TILES=256,512,5" "FLUID_TB_TAIL_TILES=256,512,7" 2>&1 | tee -a $OUT/ab_mix_tiles.txt
