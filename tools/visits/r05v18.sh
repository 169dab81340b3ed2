#!/bin/bash
# round 5, visit 18: the chained loop's counters counted up from call to call (no memset in the stream per launch): parity, then the step
OUT=$PWD/gpurun_out/r05v18; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1200 python -m pytest tests/test_jacobi_chain.py tests/test_big_passes_4096.py tests/test_long_horizon.py tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_hip_properties.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.txt
timeout 300 python tools/ab_env.py --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=0" 2>&1 | tee $OUT/chain_epoch_ab.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
FLUID_HIP_LIB=$PROBES FLUID_SKIP_CURL=1 timeout 600 python tools/overlap_vs_link.py --config stripe --quick --rounds 2 2>&1 | grep "link   0\|link  60" | tee $OUT/rank_stripe.txt
