#!/bin/bash
# round 4, visit 11: the canonical profile set on the final build
OUT=$PWD/gpurun_out/r04v11; mkdir -p $OUT gpurun_out/r04final3
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -rsx > gpurun_out/r04final3/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r04final3/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
bash tools/gpu_round.sh r04final3 quick > $OUT/gpu_round.log 2>&1; tail -12 $OUT/gpu_round.log | cut -c1-200
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-parity 2>/dev/null > $OUT/driver_flags_$i.json; python - <<PY
import json; d=json.loads(open("$OUT/driver_flags_$i.json").read().strip().splitlines()[-1]); print("driver flags run $i: %.4f ms/step  cold %.4f  steady %.4f" % (d["ms_per_step"], d.get("cold_start",{}).get("ms_per_step",0), d.get("steady_ms_per_step",0)))
PY
done
