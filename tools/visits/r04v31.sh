#!/bin/bash
# round 4, visit 31: what do the step marks (an event between the timed steps) cost the driver's 20 steps?
OUT=$PWD/gpurun_out/r04v31; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for r in 1 2 3 4; do
for f in "" "--no-step-marks"; do
echo -n "[$f] " | tee -a $OUT/marks_ab.txt
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steps_per_sec'])" | tee -a $OUT/marks_ab.txt
done; done
