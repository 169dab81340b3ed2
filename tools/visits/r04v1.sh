#!/bin/bash
# round 4, visit 1: the whole GPU suite on the new build (curl field stored by a call's last step only, step marks, schedule info),
# the first-steps probe, then interleaved A/Bs at 4096^2: curl store skip on / off, the two-texel tile as head / tail
# of the mixed Jacobi launch, and the driver's own flags with the new `timed_window_regime`.
OUT=gpurun_out/r04v1; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -x -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python tools/first_steps.py > $OUT/first_steps_raw.txt 2> $OUT/first_steps.err; tail -3 $OUT/first_steps.err; head -30 $OUT/first_steps_raw.txt
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "" "FLUID_SKIP_CURL=0" "FLUID_TB_TAIL_TILES=192,384,2" "FLUID_TB_TAIL_TILES=384,768,2" "FLUID_TB_TAIL_TILES=576,768,2" "FLUID_TB_TAIL_TILES=0,768,2" > $OUT/ab_4096.txt 2>&1; cat $OUT/ab_4096.txt
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-steady --no-parity 2>/dev/null > $OUT/driver_flags_$i.json
python - <<PY
import json; d=json.loads(open("$OUT/driver_flags_$i.json").read().strip().splitlines()[-1]); print("driver flags run $i: %.4f ms/step" % d["ms_per_step"], d.get("timed_window_regime",{}).get("ms_per_timed_step"))
PY
done
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-budget 0 > $OUT/bench_driver_flags_full.json 2> $OUT/bench.err; cut -c1-1500 $OUT/bench_driver_flags_full.json
