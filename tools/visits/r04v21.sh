#!/bin/bash
# round 4, visit 21: the dye != sim advection with its velocity taps from a wave-private LDS run: parity of everything that
# runs a dye grid != sim grid, then A/B (lab knob FLUID_VTILE) at sim 1024 / dye 4096 and at the shipping shape sim 128 / dye 1024
OUT=$PWD/gpurun_out/r04v21; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -rsx -x > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 600 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 3 "FLUID_VTILE=1" "FLUID_VTILE=0" > $OUT/ab_vtile_1024_4096.txt 2>&1; cat $OUT/ab_vtile_1024_4096.txt | cut -c1-260
timeout 600 python tools/ab_passes.py --sim 128 --dye 1024 --iters 20 --rounds 3 "FLUID_VTILE=1" "FLUID_VTILE=0" > $OUT/ab_vtile_128_1024.txt 2>&1; cat $OUT/ab_vtile_128_1024.txt | cut -c1-260
timeout 600 python tools/ab_passes.py --sim 512 --dye 2048 --iters 20 --rounds 2 "FLUID_VTILE=1" "FLUID_VTILE=0" > $OUT/ab_vtile_512_2048.txt 2>&1; cat $OUT/ab_vtile_512_2048.txt | cut -c1-260
