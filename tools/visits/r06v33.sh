#!/bin/bash
# round 6, visit 33: K6 as one more block of the chained pressure launch (lab: FLUID_CHAIN_GS=1) — parity, then the whole step A/B, then its kernels
OUT=$PWD/gpurun_out/r06v33; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 900 python tools/chain_gs_check.py 2>&1 | tee $OUT/check.txt
timeout 900 python tools/ab_env.py --rounds 4 --args "--steps 200 --warmup 50 --no-profile-pass --no-parity" "FLUID_CHAIN_GS=0" "FLUID_CHAIN_GS=1" 2>&1 | tee $OUT/ab_4096.txt
timeout 600 python tools/ab_env.py --rounds 3 --args "--size 3072 --steps 200 --warmup 50 --no-profile-pass --no-parity" "FLUID_CHAIN_GS=0" "FLUID_CHAIN_GS=1" 2>&1 | tee $OUT/ab_3072.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--iters 200 --steps 60 --warmup 20 --no-profile-pass --no-parity" "FLUID_CHAIN_GS=0" "FLUID_CHAIN_GS=1" 2>&1 | tee $OUT/ab_4096_200.txt
( cd /tmp && FLUID_HIP_LIB=$GRAFT_REPO_ROOT/webgl-fluid-simulation_amd/libfluid_hip_probes.so FLUID_CHAIN_GS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o ks -- \
    python "$GRAFT_REPO_ROOT/bench.py" --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass --settle-ms 0 >"$OUT/bench_under_rocprof.json" 2>"$OUT/rocprof.err" )
KS=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats_gs.csv" && head -8 "$OUT/kernel_stats_gs.csv" | cut -c1-100,250-420
rm -rf "$OUT/prof"
