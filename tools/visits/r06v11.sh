#!/bin/bash
# round 6, visit 11: the chained launch in PANELS (a tile row longer than an XCD holds walks band x panel rectangles) and with up to 24 blocks:
# bitwise at the widths round 5's rule left to plain launches, then A/B against the plain launches and against round 5's one-panel order
OUT=$PWD/gpurun_out/r06v11; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== correctness (product rule, then the lab build chaining everywhere) =="
timeout 900 python tools/chain_check.py "" 2>&1 | tee $OUT/chain_check.txt
timeout 1500 python tools/chain_check.py --shapes "8192x8192x50 6144x6144x50 3072x3072x47 16384x2048x200 5000x3000x33 9000x2100x21 2048x2048x50" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_PANEL=7" 2>&1 | tee -a $OUT/chain_check.txt
echo "== A/B: plain launches / chained in panels / chained, one panel (round 5's order) =="
for sz in 8192 6144; do
  timeout 900 python tools/ab_env.py --rounds 2 --args "--size $sz --steps 40 --warmup 10 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_PANEL=0" 2>&1 | sed "s/^/[$sz] /" | tee -a $OUT/panel_ab.txt
done
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 6144 --steps 40 --warmup 10 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_PANEL=9" 2>&1 | sed "s/^/[6144] /" | tee -a $OUT/panel_ab.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 3072 --steps 100 --warmup 30 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_PANEL=7" 2>&1 | sed "s/^/[3072] /" | tee -a $OUT/panel_ab.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 16384 --iters 200 --steps 6 --warmup 2 --no-profile-pass --no-parity" "FLUID_JACOBI_CHAIN=0" "FLUID_JACOBI_CHAIN=1" 2>&1 | sed "s/^/[16384 200] /" | tee -a $OUT/panel_ab.txt
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for cfg in deep16 deep; do
  for ch in 0 1; do
    echo "== one stripe rank alone ($cfg), FLUID_JACOBI_CHAIN=$ch =="
    FLUID_HIP_LIB=$PROBES FLUID_JACOBI_CHAIN=$ch timeout 600 python tools/overlap_vs_link.py --config $cfg --quick --rounds 2 2>&1 | grep "link   0\|link  60" | tee -a $OUT/rank_${cfg}_chain$ch.txt
  done
done
