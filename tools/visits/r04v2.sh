#!/bin/bash
# round 4, visit 2: the product / lab split on the GPU (knob tests through libfluid_hip_probes.so), the per-frame path that works ahead, the
# compressed code-object build, bench.py with --settle-ms under the driver's flags, the gradient-subtract fold at 4096^2 once more, and the
# interior-first overlap against a link that takes time.
OUT=gpurun_out/r04v2; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_properties.py tests/test_bench_live.py tests/test_stripes_gpu.py -m gpu -x -q -rsx > $OUT/pytest_subset.txt 2>&1; tail -4 $OUT/pytest_subset.txt
FLUID_HIP_LIB=$PWD/build_ab/compress/libfluid_hip.so timeout 300 python -m pytest tests/test_hip_vs_golden.py -m gpu -x -q > $OUT/pytest_compressed_lib.txt 2>&1; tail -2 $OUT/pytest_compressed_lib.txt
timeout 600 python tools/bench_single_step.py > $OUT/single_step.txt 2>&1; cat $OUT/single_step.txt
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-parity 2>/dev/null > $OUT/driver_flags_settle_$i.json
python - <<PY
import json; d=json.loads(open("$OUT/driver_flags_settle_$i.json").read().strip().splitlines()[-1]); print("driver flags + settle, run $i: %.4f ms/step  cold %.4f  steady %.4f" % (d["ms_per_step"], d.get("cold_start",{}).get("ms_per_step",0), d.get("steady_ms_per_step",0)), d.get("timed_window_regime",{}).get("ms_per_timed_step"))
PY
done
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_SKIP_CURL=1" "FLUID_FOLD_GRADSUB=1" > $OUT/ab_fold_4096.txt 2>&1; cat $OUT/ab_fold_4096.txt
timeout 1500 python tools/overlap_vs_link.py --rounds 1 > $OUT/overlap_vs_link_latency.txt 2>&1; cat $OUT/overlap_vs_link_latency.txt
