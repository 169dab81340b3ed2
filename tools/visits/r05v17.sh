#!/bin/bash
# round 5, visit 17: the chained pressure loop on stripe / tile ranks (the launches behind the cut ones, a row / column range per block):
# parity of every decomposition test, then one rank alone on the GPU with and without it.
OUT=$PWD/gpurun_out/r05v17; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1200 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_long_horizon.py tests/test_jacobi_chain.py tests/test_hip_f16.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for cfg in stripe tile; do
  echo "== one rank alone ($cfg): chained loop (default) =="
  FLUID_HIP_LIB=$PROBES FLUID_SKIP_CURL=1 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | grep "link   0\|link  60\|40 GB" | tee $OUT/rank_${cfg}_chain.txt
  echo "== one rank alone ($cfg): FLUID_JACOBI_CHAIN=0 =="
  FLUID_HIP_LIB=$PROBES FLUID_JACOBI_CHAIN=0 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | grep "link   0\|link  60\|40 GB" | tee $OUT/rank_${cfg}_nochain.txt
done
echo "== the single domain, same box =="
timeout 300 python tools/ab_env.py --rounds 1 "FLUID_SKIP_CURL=1" 2>&1 | tee $OUT/single_4096.txt
echo "== in-process groups =="
timeout 300 python tools/bench_group.py 4096 50 56 4 1 2>&1 | tail -1 | tee $OUT/group.txt
timeout 300 python tools/bench_group.py 4096 50 56 4 2 2>&1 | tail -1 | tee -a $OUT/group.txt
