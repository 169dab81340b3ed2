#!/bin/bash
# One GPU-box visit: parity tests, smoke, headline bench, rocprofv3 kernel stats, PMC traffic passes.
# Usage (from the repo root, on the GPU box via gpurun):  bash tools/gpu_round.sh <tag> [quick]
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-run}
MODE=${2:-full}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1

echo "== build check (prebuilt .so must load) ==" | tee "$OUT/log.txt"
python -c "import sys; sys.path.insert(0,'webgl-fluid-simulation_amd'); import fluid_hip; fluid_hip.lib(); print('libfluid_hip ok')" >>"$OUT/log.txt" 2>&1

if [ "$MODE" != "quick" ]; then
  echo "== pytest -m gpu ==" | tee -a "$OUT/log.txt"
  timeout 1500 python -m pytest tests -m gpu -x -q -rsx >"$OUT/pytest_gpu.txt" 2>&1
  echo "pytest exit $?" | tee -a "$OUT/log.txt"
  tail -5 "$OUT/pytest_gpu.txt" | tee -a "$OUT/log.txt"

  echo "== smoke ==" | tee -a "$OUT/log.txt"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >>"$OUT/log.txt" 2>&1
  echo "smoke exit $?" | tee -a "$OUT/log.txt"
fi

echo "== bench (headline; the PMC traffic passes and the reference baseline run inside it) ==" | tee -a "$OUT/log.txt"
FLUID_BENCH_KEEP_PMC="$OUT" timeout 1200 python bench.py >"$OUT/bench.json" 2>"$OUT/bench.err"
echo "bench exit $?" | tee -a "$OUT/log.txt"
cat "$OUT/bench.json" | tee -a "$OUT/log.txt"

echo "== bench as the driver runs it: --steps 20 --warmup 5 ==" | tee -a "$OUT/log.txt"
timeout 900 python bench.py --steps 20 --warmup 5 >"$OUT/bench_driver_flags.json" 2>>"$OUT/bench.err"
cat "$OUT/bench_driver_flags.json" | tee -a "$OUT/log.txt"

echo "== bench passes schedule ==" | tee -a "$OUT/log.txt"
FLUID_BENCH_KEEP_PMC="$OUT" timeout 900 python bench.py --schedule passes --cpu-budget 0 --no-steady >"$OUT/bench_passes.json" 2>>"$OUT/bench.err"
cat "$OUT/bench_passes.json" | tee -a "$OUT/log.txt"

echo "== bench --storage f16 (side mode, SURVEY 8f N4) ==" | tee -a "$OUT/log.txt"
timeout 600 python bench.py --storage f16 --cpu-budget 0 --no-traffic --no-steady >"$OUT/bench_f16.json" 2>>"$OUT/bench.err"
cat "$OUT/bench_f16.json" | tee -a "$OUT/log.txt"

echo "== rocprofv3 kernel trace ==" | tee -a "$OUT/log.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o ks -- \
    python "$GRAFT_REPO_ROOT/bench.py" --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass --settle-ms 0 >"$OUT/bench_under_rocprof.json" 2>"$OUT/rocprof.err" )
echo "rocprof exit $?" | tee -a "$OUT/log.txt"
KS=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats.csv" && head -12 "$OUT/kernel_stats.csv" | cut -c1-200 | tee -a "$OUT/log.txt"
# keep the merged-back payload small
rm -rf "$OUT/prof"
echo "== the display compositor at the shipping sizes: timing, then its kernels under rocprofv3 ==" | tee -a "$OUT/log.txt"
timeout 300 python tools/bench_render.py >"$OUT/bench_render.json" 2>>"$OUT/bench.err"; cat "$OUT/bench_render.json" | tee -a "$OUT/log.txt"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_render" -o rs -- python "$GRAFT_REPO_ROOT/tools/bench_render.py" 100 >/dev/null 2>>"$OUT/rocprof.err" )
KS=$(find "$OUT/prof_render" -name '*kernel_stats.csv' | head -1)
[ -n "$KS" ] && cp "$KS" "$OUT/render_kernel_stats.csv" && head -16 "$OUT/render_kernel_stats.csv" | cut -c1-160 | tee -a "$OUT/log.txt"
rm -rf "$OUT/prof_render"
echo "== other sizes, shipping defaults, SQ counters ==" | tee -a "$OUT/log.txt"
bash tools/other_sizes.sh "$TAG" >>"$OUT/log.txt" 2>&1
timeout 600 python tools/bench_shipping.py >"$OUT/bench_shipping_defaults.json" 2>>"$OUT/bench.err"
bash tools/pmc_sq.sh "$TAG/sq" >/dev/null 2>&1; cp "$OUT/sq/sq_summary.txt" "$OUT/sq_counters_step_kernels.txt" 2>/dev/null
echo "== done ==" | tee -a "$OUT/log.txt"
