#!/bin/bash
# round 6, visit 7: the chained launch's edge waves skip the rows the apron has reached, light blocks on waves 0 and 2 (one light wave per SIMD)
OUT=$PWD/gpurun_out/r06v7; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 600 python tools/chain_check.py "FLUID_CHAIN_SKIP=1" "" 2>&1 | tee $OUT/chain_check.txt
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 100 --warmup 30 --no-profile-pass" "FLUID_CHAIN_SKIP=0" "FLUID_CHAIN_SKIP=1" 2>&1 | tee $OUT/skip_ab.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/driver_cmd.json 2>>$OUT/bench.err; python - $OUT/driver_cmd.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd (product): ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "roofline", {k: (d.get("roofline") or {}).get(k) for k in ("frac", "avg_launch_ms", "traffic")}, "err", d.get("error"))
PY
