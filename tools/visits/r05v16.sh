#!/bin/bash
# round 5, visit 16: the chained Jacobi launch as the PRODUCT path on 4096-wide grids — its tests, the suites that run big grids, the bench line
OUT=$PWD/gpurun_out/r05v16; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python -m pytest tests/test_jacobi_chain.py tests/test_big_passes_4096.py tests/test_long_horizon.py tests/test_bench_live.py tests/test_device_view.py tests/test_hip_properties.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest.txt
timeout 400 python bench.py > $OUT/bench.json 2>$OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['steps_per_sec'], d['value'], r['kernel'], r['frac'], r['traffic'], r['avg_launch_ms'], r['blocks_per_dispatch'], r['frac_compulsory'], d['step_hbm']['frac'], d['config']['kernels'], d['pass_ms_per_step'], r.get('valu',{}).get('busy_frac_issue_cost'), r.get('valu',{}).get('effective_clock_GHz'), d['parity_in_run']['ok'])"
timeout 300 python tools/ab_env.py --rounds 2 "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=0" 2>&1 | tee $OUT/chain_product_ab.txt
