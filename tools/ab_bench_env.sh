#!/bin/bash
# On the GPU box: headline bench under different values of one environment knob; prints steps/s and per-pass ms.
# Usage: bash tools/ab_bench_env.sh <tag> <ENVVAR> "<values>"
TAG=$1; VAR=$2; VALS=$3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for V in $VALS; do
  env $VAR=$V python bench.py --steps 60 --warmup 10 --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('$VAR=$V steps/s %.1f  ms/step %.4f  passes %s' % (d['steps_per_sec'], d['ms_per_step'], {k:v for k,v in p.items() if v}))" | tee -a $OUT/ab.txt
done
