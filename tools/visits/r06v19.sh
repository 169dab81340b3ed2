#!/bin/bash
# round 6, visit 19: the fused advection walking its block rows from the last to the first (what the gradient subtract in front of it wrote last is
# what the Infinity Cache still holds) — lab kernel k_advect_both_fast_rgb_wy<4, 1>: FLUID_ADVECT_XCD=4 plain, =2 reversed
OUT=$PWD/gpurun_out/r06v19; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 100 --warmup 30" "FLUID_ADVECT_XCD=4" "FLUID_ADVECT_XCD=2" 2>&1 | tee $OUT/advect_reverse_ab.txt
echo "== fluid_set_curl_output =="
timeout 900 python -m pytest tests/test_curl_output.py tests/test_abi.py tests/test_node_shim.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_curl_output.txt
timeout 600 python bench.py --cpu-budget 0 --no-traffic > $OUT/bench_per_frame.json 2>$OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_per_frame.json')); print(d['ms_per_step'], d.get('steady_ms_per_step'), json.dumps(d.get('per_frame'))[:600])"
