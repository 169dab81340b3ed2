#!/bin/bash
# round 5, visit 6: visit 5's dye != sim part again — its libraries were stale (the typed-LDS-pointer source did not compile on the host side and
# the build step's output was not looked at: the library that ran still had the flat loads, and the box variant read its box through them).
OUT=$PWD/gpurun_out/r05v6; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== parity of everything that advects on two grids =="
timeout 1200 python -m pytest tests/test_hip_vs_golden.py tests/test_hip_vs_oracle.py tests/test_hip_properties.py tests/test_long_horizon.py tests/test_input_replay.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.txt
echo "== sim 1024 / dye 4096 / 20 =="
timeout 600 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_DYE_BOX=1" "FLUID_DYE_PACK=0" 2>&1 | tee $OUT/dye_ne_sim_1024_4096.txt
echo "== sim 2048 / dye 8192 / 20 =="
timeout 600 python tools/ab_passes.py --sim 2048 --dye 8192 --iters 20 --rounds 2 --steps 200 "FLUID_SKIP_CURL=1" "FLUID_DYE_BOX=1" 2>&1 | tee $OUT/dye_ne_sim_2048_8192.txt
echo "== sim 256 / dye 2048 / 20 (the reference's ratio 8) =="
timeout 600 python tools/ab_passes.py --sim 256 --dye 2048 --iters 20 --rounds 2 "FLUID_SKIP_CURL=1" 2>&1 | tee $OUT/dye_ne_sim_256_2048.txt
echo "== the reference's shipping configuration (sim 128 / dye 1024 / 20) =="
timeout 300 python tools/bench_shipping.py > $OUT/bench_shipping_defaults.json 2>&1; python - <<P
import json
t=open('$OUT/bench_shipping_defaults.json').read(); d=json.loads(t[t.index('{'):])
for c in d['kernels_fast']: print(c['case'], c.get('throughput_us_per_step'), c.get('latency_us_per_step_median'), c['pass_us_per_step'])
P
