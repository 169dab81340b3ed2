#!/bin/bash
# round 6, visit 23: the rotation as the PRODUCT default with the new rule (3072^2 ... 20 M texels, any width): parity, then the step at the sizes it newly covers
OUT=$PWD/gpurun_out/r06v23; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1700 python -m pytest tests/test_jacobi_chain.py tests/test_chain_safety.py tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_long_horizon.py tests/test_big_passes_4096.py tests/test_hip_properties.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest.txt
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 100 --warmup 30 --no-profile-pass" "" "FLUID_CHAIN_ROT=0" "FLUID_JACOBI_CHAIN=0" 2>&1 | tee $OUT/step_4096.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 3072 --steps 150 --warmup 40 --no-profile-pass --no-parity" "" "FLUID_JACOBI_CHAIN=0" 2>&1 | sed "s/^/[3072] /" | tee $OUT/step_3072.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for cfg in stripe deep; do
  for st in "" "FLUID_CHAIN_ROT=0" "FLUID_JACOBI_CHAIN=0"; do
    echo "== one rank alone ($cfg) [$st] =="
    env FLUID_HIP_LIB=$PROBES $st timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | grep "link   0" | tee -a $OUT/rank_${cfg}.txt
  done
done
