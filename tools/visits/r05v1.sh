#!/bin/bash
# round 5, visit 1: prove the diagnosis of BENCH_r04's MISMATCH (device_view race on the packed dye's conversion) on round 4's own tree,
# then show the fixed contract green: the race tool, the new GPU tests, and the driver's exact command seven times in a row.
OUT=$PWD/gpurun_out/r05v1; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "old: $(sha256sum build_ab/r04tree/webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  new: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)"
echo "== race tool, round 4's tree (git archive d63ebbf) =="
timeout 300 python tools/device_view_race.py --tree build_ab/r04tree --trials 30 --modes legacy,unsynced > $OUT/race_r04tree.json 2>$OUT/race_r04tree.err; echo "exit $?"; cat $OUT/race_r04tree.json
echo "== round 4's bench.py, its own command, 6 times =="
for i in 1 2 3 4 5 6; do (cd build_ab/r04tree && timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --cpu-budget 0 --no-steady 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 bench run $i:', d.get('ms_per_step'), (d.get('error') or 'ok')[:200])"); done | tee $OUT/r04_bench_runs.txt
echo "== race tool, this tree =="
timeout 400 python tools/device_view_race.py --trials 50 > $OUT/race_r05.json 2>$OUT/race_r05.err; echo "exit $?"; cat $OUT/race_r05.json
echo "== the new GPU tests =="
timeout 900 python -m pytest tests/test_device_view.py tests/test_bench_live.py -x -q -m gpu > $OUT/pytest_new.txt 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_new.txt
echo "== the driver's command, 7 times =="
for i in 1 2 3 4 5 6 7; do timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2>$OUT/driver_cmd_$i.err; echo "run $i exit $?"; python -c "
import json; d=json.load(open('$OUT/driver_cmd_$i.json')); print(d.get('ms_per_step'), d.get('value'), d.get('roofline',{}).get('frac'), d.get('parity_in_run',{}).get('ok'), 'cpu_baseline' in d, d.get('preloaded_window',{}).get('ms_per_step'), d.get('error'))"; done
