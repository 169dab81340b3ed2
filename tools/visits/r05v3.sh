#!/bin/bash
# round 5, visit 3: the stripe / tile rank after this round's two changes — the dye packed on ranks of >= 3072^2 texels (ghost texels travel as
# 12-byte texels) and the thin launches behind an exchange (strips, Jacobi frames) on the comm stream beside the interiors.
# Parity first (every stripe / tile test incl. the BASELINE sizes), then one rank alone on the GPU (tests/fake_rccl loopback) against round 4's schedule.
OUT=$PWD/gpurun_out/r05v3; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== stripe / tile parity =="
timeout 1500 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_long_horizon.py tests/test_hip_f16.py tests/test_hip_properties.py tests/test_node_shim.py -m gpu -x -q > $OUT/pytest_stripes.txt 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest_stripes.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for cfg in stripe tile; do
  echo "== one rank alone ($cfg): round 5 schedule (lab build, defaults) =="
  FLUID_HIP_LIB=$PROBES FLUID_SKIP_CURL=1 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | tee $OUT/rank_${cfg}_r05.txt
  echo "== one rank alone ($cfg): strips and frames on the context stream again (FLUID_STRIPS_ON_COMM=0) =="
  FLUID_HIP_LIB=$PROBES FLUID_STRIPS_ON_COMM=0 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | tee $OUT/rank_${cfg}_strips_main.txt
  echo "== one rank alone ($cfg): round 4's schedule (FLUID_STRIPS_ON_COMM=0 FLUID_DYE_PACK=0) =="
  FLUID_HIP_LIB=$PROBES FLUID_STRIPS_ON_COMM=0 FLUID_DYE_PACK=0 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | tee $OUT/rank_${cfg}_r04.txt
done
echo "== the single domain of the same size, same box (bench.py 100 steps, lab build) =="
timeout 300 python tools/ab_env.py --rounds 2 "FLUID_SKIP_CURL=1" 2>&1 | tee $OUT/single_4096.txt
