#!/bin/bash
# round 6, visit 32: the plan's pressure blocks balanced in whole launches (200 iterations at halo 56: 4 x 50 instead of 53 + 53 + 53 + 41 = 23 launches): the
# 200-iteration ranks, then the decomposition tests
OUT=$PWD/gpurun_out/r06v32; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)"
for cfg in deep16 deep stripe; do
  echo "== one rank alone ($cfg), product library =="
  timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | grep "link   0\|link  60" | tee -a $OUT/rank_${cfg}.txt
done
timeout 1700 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_long_horizon.py tests/test_jacobi_chain.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.txt
