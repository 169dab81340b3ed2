#!/bin/bash
# round 4, visit 7: the dye packed to three floats per texel inside the big fused advection: the whole GPU suite, the A/B against the RGBA
# path (lab knob FLUID_DYE_PACK=0), the driver's flags
OUT=$PWD/gpurun_out/r04v7; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -rsx -x > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 200 --warmup 50 --no-parity" "FLUID_SKIP_CURL=1" "FLUID_DYE_PACK=0" > $OUT/ab_dye_pack_4096.txt 2>&1; cat $OUT/ab_dye_pack_4096.txt
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 8192 --steps 60 --warmup 20 --no-parity" "FLUID_SKIP_CURL=1" "FLUID_DYE_PACK=0" > $OUT/ab_dye_pack_8192.txt 2>&1; cat $OUT/ab_dye_pack_8192.txt
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-traffic --no-parity 2>/dev/null > $OUT/driver_flags_$i.json
python - <<PY
import json; d=json.loads(open("$OUT/driver_flags_$i.json").read().strip().splitlines()[-1]); print("driver flags run $i: %.4f ms/step  cold %.4f  steady %.4f  packed %s" % (d["ms_per_step"], d.get("cold_start",{}).get("ms_per_step",0), d.get("steady_ms_per_step",0), d["config"]["kernels"].get("dye_packed_rgb")), d.get("pass_ms_per_step"))
PY
done
