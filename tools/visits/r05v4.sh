#!/bin/bash
# round 5, visit 4: the stripe / tile rank with ev_joined at device scope (one system-scope fence fewer on the chain behind an exchange), against
# round 4's schedule, fast and slow links; the whole set as an in-process group against the single domain (round 4's decomposition_overhead table).
OUT=$PWD/gpurun_out/r05v4; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== parity: the rank-thread and loopback tests (RCCL path of the driver) =="
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py -m gpu -x -q > $OUT/pytest_stripes.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_stripes.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for cfg in stripe tile deep; do
  echo "== one rank alone ($cfg): round 5 =="
  FLUID_HIP_LIB=$PROBES FLUID_SKIP_CURL=1 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | tee $OUT/rank_${cfg}_r05.txt
  echo "== one rank alone ($cfg): round 4's schedule =="
  FLUID_HIP_LIB=$PROBES FLUID_STRIPS_ON_COMM=0 FLUID_DYE_PACK=0 timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | tee $OUT/rank_${cfg}_r04.txt
done
echo "== in-process groups against the single domain =="
for t in 1 2; do timeout 600 python tools/bench_group.py 4096 50 56 4 $t 2>&1 | tail -3 | tee -a $OUT/group_vs_single.txt; done
timeout 300 python tools/bench_group.py 4096 50 56 2 1 2>&1 | tail -2 | tee -a $OUT/group_vs_single.txt
echo "== the single domain, same box =="
timeout 300 python tools/ab_env.py --rounds 2 "FLUID_SKIP_CURL=1" 2>&1 | tee $OUT/single_4096.txt
