#!/bin/bash
# round 2, visit 8: same-box A/B of the pitch refactor (old checkout under build_ab/old_repo) and a repeat of the node tile-rank test
set -u
OUT=$PWD/gpurun_out/r02_v8; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('%-10s steps/s %7.1f  ms/step %.4f  cvd %.4f jacobi %.4f gradsub %.4f advect %.4f' % ('$1', d['steps_per_sec'], d['ms_per_step'], p['vorticity_ms'], p['jacobi_ms'], p['gradsub_ms'], p['advect_dye_ms']))"; }
for rep in 1 2 3; do
( cd build_ab/old_repo && python bench.py --steps 150 --warmup 50 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null ) | line old | tee -a $OUT/log.txt
python bench.py --steps 150 --warmup 50 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null | line new | tee -a $OUT/log.txt
done
( cd build_ab/old_repo && python bench.py --schedule passes --steps 60 --warmup 20 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null ) | line old-passes | tee -a $OUT/log.txt
python bench.py --schedule passes --steps 60 --warmup 20 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null | line new-passes | tee -a $OUT/log.txt
echo "== node tile rank test, five times ==" | tee -a $OUT/log.txt
for k in 1 2 3 4 5; do
timeout 200 python -m pytest tests/test_node_shim.py -m gpu -x -q -k "tile_rank or launcher" 2>&1 | tail -1 | tee -a $OUT/log.txt
done
echo "== hip vs golden (incl. the subnormal fixture) ==" | tee -a $OUT/log.txt
timeout 600 python -m pytest tests/test_hip_vs_golden.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
