#!/bin/bash
# round 5, visit 2 (second lease): the whole GPU suite on the ABI-9 library, the driver's exact command seven more times, and two cheap A/Bs
# through the lab build: the packed dye's two taps of a row as dwordx4 + dwordx2 (FLUID_RGB_PAIR=1) and the 13-deep Jacobi shape (4 launches).
OUT=$PWD/gpurun_out/r05v2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q -rsx > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest_gpu.txt
echo "== the driver's command, 7 times =="
for i in 1 2 3 4 5 6 7; do timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2>$OUT/driver_cmd_$i.err; echo "run $i exit $?"; python -c "
import json; d=json.load(open('$OUT/driver_cmd_$i.json')); print(d.get('ms_per_step'), d.get('value'), d.get('roofline',{}).get('frac'), d.get('parity_in_run',{}).get('ok'), 'cpu_baseline' in d, d.get('preloaded_window',{}).get('ms_per_step'), d.get('error'))"; done
echo "== A/B: paired packed-dye taps, 4096^2 =="
timeout 600 python tools/ab_env.py --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_RGB_PAIR=1" 2>&1 | tee $OUT/rgb_pair_ab_4096.txt
echo "== A/B: paired packed-dye taps, sim 1024 / dye 4096 =="
timeout 600 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_RGB_PAIR=1" 2>&1 | tee $OUT/rgb_pair_ab_dye_ne_sim.txt
echo "== A/B: Jacobi 13 deep (shape 4: 4 launches of 13/13/12/12) against the shipped 10 deep =="
timeout 600 python tools/ab_env.py --rounds 2 "FLUID_SKIP_CURL=1" "FLUID_TB_VARIANT=4" "FLUID_TB_VARIANT=7" 2>&1 | tee $OUT/jacobi_depth13_ab.txt
