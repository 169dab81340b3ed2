#!/bin/bash
# Copies what one tools/gpu_round.sh visit produced (gpurun_out/<tag>/) into the tracked profiles/<round>/ files.
# Usage: bash tools/publish_profiles.sh <tag> [round dir, default r01]
set -e
F=gpurun_out/$1; R=profiles/${2:-r01}
cp $F/bench.json $R/bench_4096_50.json
cp $F/bench_passes.json $R/bench_4096_50_passes_schedule.json
cp $F/bench_stripes1.json $R/bench_stripes_driver_n1.json
cp $F/bench_f16.json $R/bench_4096_50_f16_storage.json
cp $F/bench_under_rocprof.json $R/bench_under_rocprof.json
cp $F/kernel_stats.csv $R/kernel_stats_fused_4096_50.csv
cp $F/pmc_FETCH_SIZE_fused.csv $R/pmc_fetch_fused.csv
cp $F/pmc_FETCH_SIZE_passes.csv $R/pmc_fetch_passes.csv
cp $F/pmc_WRITE_SIZE_fused.csv $R/pmc_write_fused.csv
cp $F/pmc_WRITE_SIZE_passes.csv $R/pmc_write_passes.csv
cp $F/traffic.json $R/traffic_4096_50.json
cp $F/traffic.json profiles/traffic_latest.json
grep -h "passed" $F/pytest_gpu.txt | tail -1 > $R/pytest_gpu.txt
echo "published $F -> $R"
