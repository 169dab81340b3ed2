#!/bin/bash
# round 5, visit 19 (the last): the FINAL library — the whole GPU suite, smoke, the driver's command five times, the default bench line with its
# PMC passes kept, the bench under rocprofv3 --kernel-trace --stats.
OUT=$PWD/gpurun_out/r05v19; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  bench.py: $(sha256sum bench.py | cut -c1-16)"
timeout 1500 python -m pytest tests -m gpu -q -rsx > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3 4 5; do timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2>$OUT/driver_cmd_$i.err; echo "run $i exit $?"; python -c "
import json; d=json.load(open('$OUT/driver_cmd_$i.json')); print(d.get('ms_per_step'), d.get('value'), d.get('roofline',{}).get('frac'), d.get('parity_in_run',{}).get('ok'), 'cpu_baseline' in d, d.get('preloaded_window',{}).get('ms_per_step'), d.get('error'))"; done
FLUID_BENCH_KEEP_PMC="$OUT" timeout 600 python bench.py > $OUT/bench.json 2>$OUT/bench.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print(d['ms_per_step'], d['steps_per_sec'], d['value'], r['kernel'], r['frac'], r['traffic'], r['avg_launch_ms'], r['frac_compulsory'], d['step_hbm']['frac'], d['pass_ms_per_step']['jacobi_ms'], d['parity_in_run']['ok'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o ks -- python "$GRAFT_REPO_ROOT/bench.py" --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass >"$OUT/bench_under_rocprof.json" 2>"$OUT/rocprof.err" ); KS=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1); [ -n "$KS" ] && cp "$KS" "$OUT/kernel_stats.csv" && head -6 "$OUT/kernel_stats.csv" | cut -c1-60,300-420; rm -rf "$OUT/prof"
