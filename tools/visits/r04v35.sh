#!/bin/bash
# round 4, visit 35: the stripe driver's events without a system-scope fence (lab: FLUID_EVENT_SCOPE=device) — the loopback stripe rank, A/B
OUT=$PWD/gpurun_out/r04v35; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for r in 1 2 3; do
  echo "== system scope (default), round $r" | tee -a $OUT/event_scope_ab.txt
  FLUID_HIP_LIB=$P FLUID_SKIP_CURL=1 timeout 200 python tools/overlap_vs_link.py --quick --rounds 1 --config stripe 2>&1 | grep "ms/step" | cut -c1-70 | tee -a $OUT/event_scope_ab.txt
  echo "== FLUID_EVENT_SCOPE=device, round $r" | tee -a $OUT/event_scope_ab.txt
  FLUID_HIP_LIB=$P FLUID_EVENT_SCOPE=device timeout 200 python tools/overlap_vs_link.py --quick --rounds 1 --config stripe 2>&1 | grep "ms/step" | cut -c1-70 | tee -a $OUT/event_scope_ab.txt
done
