#!/bin/bash
# Copies what one tools/gpu_round.sh visit produced (gpurun_out/<tag>/) into the tracked profiles/<round>/ files.
# Usage: bash tools/publish_profiles.sh <tag> [round dir, default r02]
set -e
F=gpurun_out/$1; R=profiles/${2:-r03}
mkdir -p $R
cp $F/bench.json $R/bench_4096_50.json
cp $F/bench_driver_flags.json $R/bench_4096_50_steps20_warmup5.json
cp $F/bench_passes.json $R/bench_4096_50_passes_schedule.json
cp $F/bench_f16.json $R/bench_4096_50_f16_storage.json
cp $F/bench_under_rocprof.json $R/bench_under_rocprof.json
cp $F/kernel_stats.csv $R/kernel_stats_fused_4096_50.csv
# the raw PMC passes bench.py ran inside itself (FLUID_BENCH_KEEP_PMC): FETCH_SIZE / WRITE_SIZE, separate runs
for s in fused passes; do
  [ -f $F/pmc_FETCH_SIZE_$s.csv ] && cp $F/pmc_FETCH_SIZE_$s.csv $R/pmc_fetch_$s.csv
  [ -f $F/pmc_WRITE_SIZE_$s.csv ] && cp $F/pmc_WRITE_SIZE_$s.csv $R/pmc_write_$s.csv
done
for f in $F/pmc_SQ_*_fused.csv; do [ -f "$f" ] && cp "$f" $R/pmc_sq_valu_fused.csv; done
for f in bench_shipping_defaults.json bench_other_sizes.txt soak.txt sq_counters_step_kernels.txt bench_render.json render_kernel_stats.csv; do [ -f $F/$f ] && cp $F/$f $R/$f; done
tail -15 $F/pytest_gpu.txt > $R/pytest_gpu.txt
echo "published $F -> $R"
