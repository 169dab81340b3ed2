#!/bin/bash
# round 2, visit 4: streaming Jacobi with the cooperative input-row ring
set -u
OUT=$PWD/gpurun_out/r02_v4; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== bitwise tests (streaming kernel is the default) ==" | tee $OUT/log.txt
timeout 900 python -m pytest tests/test_hip_vs_oracle.py tests/test_hip_properties.py -m gpu -x -q 2>&1 | tail -5 | tee -a $OUT/log.txt
for v in 2 3; do
FLUID_JACOBI_STREAM=$v timeout 900 python -m pytest tests/test_hip_vs_oracle.py -m gpu -x -q -k jacobi 2>&1 | tail -2 | tee -a $OUT/log.txt
done
echo "== jacobi alone at 4096^2 ==" | tee -a $OUT/log.txt
for it in 1 25 50; do
for v in 1 2 3; do
for wgs in 512 768; do
FLUID_JACOBI_STREAM=$v FLUID_STREAM_WGS=$wgs TB_VARIANTS=0 timeout 120 python tools/bench_jacobi.py 4096 $it | sed "s/^/stream$v wgs$wgs /" | tee -a $OUT/log.txt
done; done; done
echo "== whole step ==" | tee -a $OUT/log.txt
for v in 0 1 2; do
FLUID_JACOBI_STREAM=$v python bench.py --steps 120 --warmup 40 --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('stream=$v steps/s %7.1f  ms/step %.4f  cvd %.4f jacobi %.4f gradsub %.4f advect %.4f' % (d['steps_per_sec'], d['ms_per_step'], p['vorticity_ms'], p['jacobi_ms'], p['gradsub_ms'], p['advect_dye_ms']))" | tee -a $OUT/log.txt
done
echo "== done ==" | tee -a $OUT/log.txt
