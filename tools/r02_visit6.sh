#!/bin/bash
# round 2, visit 6: 16-wave x 5-row tile (one workgroup alone keeps four waves per SIMD) with and without the per-CU load gate
set -u
OUT=$PWD/gpurun_out/r02_v6; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== bitwise ==" | tee $OUT/log.txt
FLUID_TB_VARIANT=8 FLUID_LOAD_GATE=1 timeout 900 python -m pytest tests/test_hip_vs_oracle.py tests/test_hip_properties.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT/log.txt
echo "== jacobi alone ==" | tee -a $OUT/log.txt
for it in 1 10 50; do for g in 0 1; do
FLUID_LOAD_GATE=$g TB_VARIANTS="0 8 9" python tools/bench_jacobi.py 4096 $it | sed "s/^/gate$g /" | tee -a $OUT/log.txt
done; done
echo "== whole step ==" | tee -a $OUT/log.txt
for v in 0 8; do for g in 0 1; do
FLUID_TB_VARIANT=$v FLUID_LOAD_GATE=$g python bench.py --steps 120 --warmup 40 --cpu-budget 0 --no-traffic --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('variant=$v gate=$g steps/s %7.1f  ms/step %.4f  cvd %.4f jacobi %.4f gradsub %.4f advect %.4f' % (d['steps_per_sec'], d['ms_per_step'], p['vorticity_ms'], p['jacobi_ms'], p['gradsub_ms'], p['advect_dye_ms']))" | tee -a $OUT/log.txt
done; done
echo "== done ==" | tee -a $OUT/log.txt
