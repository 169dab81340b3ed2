#!/bin/bash
# round 2, visit 2: per-CU load gate (anti-phase of the two resident workgroups) and nontemporal stores, A/B on the headline step
set -u
OUT=$PWD/gpurun_out/r02_v2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
run() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py --steps 120 --warmup 40 --cpu-budget 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('pass_ms_per_step',{})
print('%-22s steps/s %7.1f  ms/step %.4f  cvd %.4f jacobi %.4f gradsub %.4f advect %.4f' % ('$label', d['steps_per_sec'], d['ms_per_step'], p['vorticity_ms'], p['jacobi_ms'], p['gradsub_ms'], p['advect_dye_ms']))" | tee -a $OUT/ab.txt
}
echo "== correctness first: fused == passes bitwise with the gates on ==" | tee $OUT/log.txt
timeout 900 python -m pytest tests/test_hip_properties.py tests/test_hip_vs_oracle.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT/log.txt
for rep in 1 2; do
run "gate0" FLUID_LOAD_GATE=0
run "gate1(jacobi)" FLUID_LOAD_GATE=1
run "gate2(cvd)" FLUID_LOAD_GATE=2
run "gate3" FLUID_LOAD_GATE=3
done
for v in 1 2 6 8 72 16 127; do
run "nt$v gate0" FLUID_LOAD_GATE=0 FLUID_HIP_LIB=$PWD/build_ab/nt$v/libfluid_hip.so
done
run "nt127 gate3" FLUID_LOAD_GATE=3 FLUID_HIP_LIB=$PWD/build_ab/nt127/libfluid_hip.so
run "gate0 again" FLUID_LOAD_GATE=0
echo "== jacobi alone: 10 iterations per launch, gate off / on, and the 5x steady state ==" | tee -a $OUT/log.txt
for g in 0 1; do
FLUID_LOAD_GATE=$g TB_VARIANTS="0" python tools/bench_jacobi.py 4096 10 | sed "s/^/gate$g /" | tee -a $OUT/log.txt
FLUID_LOAD_GATE=$g TB_VARIANTS="0" python tools/bench_jacobi.py 4096 50 | sed "s/^/gate$g /" | tee -a $OUT/log.txt
FLUID_LOAD_GATE=$g FLUID_HIP_LIB=$PWD/build_ab/rep5/libfluid_hip.so TB_VARIANTS="0" python tools/bench_jacobi.py 4096 10 | sed "s/^/gate$g rep5 /" | tee -a $OUT/log.txt
FLUID_LOAD_GATE=$g TB_VARIANTS="4 5" python tools/bench_jacobi.py 4096 50 | sed "s/^/gate$g /" | tee -a $OUT/log.txt
done
cat $OUT/ab.txt >> $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
