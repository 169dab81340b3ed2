#!/bin/bash
# round 6, visit 36: does a forced performance level move the load-step clock dip (profiles/r04/first_steps.txt) the driver's window sits in?
# The driver's command under: the box as it comes / power_dpm_force_performance_level = high / perf determinism at 2400 and 2100 MHz / back to auto.
OUT=$PWD/gpurun_out/r06v36; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
run() { timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('timed_window_regime', {})
print('%-28s %.4f ms/step   first/last steps %s' % (sys.argv[1], d['ms_per_step'], str(r)[:160]))" "$1"; }
{
rocm-smi --showperflevel 2>&1 | grep -i -E "perf|level" | head -3
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>/dev/null | head -2
for k in 1 2 3; do run "as the box comes"; done
timeout 60 rocm-smi --setperflevel high 2>&1 | grep -v "^$" | head -4
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>/dev/null | head -2
for k in 1 2 3; do run "perflevel high"; done
timeout 60 rocm-smi --setperfdeterminism 2400 2>&1 | grep -v "^$" | head -4
for k in 1 2 3; do run "perf determinism 2400"; done
timeout 60 rocm-smi --setperfdeterminism 2100 2>&1 | grep -v "^$" | head -4
for k in 1 2 3; do run "perf determinism 2100"; done
timeout 60 rocm-smi --resetperfdeterminism 2>&1 | grep -v "^$" | head -3
timeout 60 rocm-smi --setperflevel auto 2>&1 | grep -v "^$" | head -3
for k in 1 2; do run "back to auto"; done
} 2>&1 | tee $OUT/perflevel.txt
