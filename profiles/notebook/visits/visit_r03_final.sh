#!/bin/bash
# the last visit of round 3: the canonical round + the driver-flag figure against the warm-up length
bash tools/gpu_round.sh r03final5
OUT=$PWD/gpurun_out/r03final5
for W in 5 10 20 40 80 160; do
  timeout 200 python bench.py --steps 20 --warmup $W --cpu-budget 0 --no-traffic --no-profile-pass --no-steady --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('--steps 20 --warmup %3d: %.4f ms/step' % ($W, d['ms_per_step']))" | tee -a $OUT/warmup_curve.txt
done
