#!/bin/bash
# round 2, visit 9: edge-mode split of the tile kernels — full GPU suite, then same-box A/B against the pre-pitch checkout
set -u
OUT=$PWD/gpurun_out/r02_v9; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/log.txt
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
( cd build_ab/old_repo && python bench.py --storage f16 --steps 150 --warmup 50 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null ) | line old-f16 | tee -a $OUT/log.txt
python bench.py --storage f16 --steps 150 --warmup 50 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null | line new-f16 | tee -a $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
