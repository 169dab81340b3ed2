#!/bin/bash
# round 5, visit 10: VERDICT r04 item 5 (i) built after all — the whole pressure loop as ONE launch of chained blocks of iterations
# (lab: FLUID_JACOBI_CHAIN=1, k_jacobi_tb_chain: per-(block, tile row) counters, write-through pressure stores, sc1 loads, odd blocks walk
# their XCD runs backwards) against the shipped five launches.  bench.py's in-run parity check (fused == per-pass, 4096^2, bitwise) runs in every line.
OUT=$PWD/gpurun_out/r05v10; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== correctness first: 6 steps at 4096^2, chain vs the per-pass schedule, every field =="
FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so FLUID_JACOBI_CHAIN=1 timeout 300 python - <<'P' 2>&1 | tail -8
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "webgl-fluid-simulation_amd"))
import numpy as np, fluid_hip
for size, iters in ((4096, 50), (4096, 47), (3200, 50)):
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    sims = [fluid_hip.FluidSim(canvas=(size, size), config=cfg, schedule=s, random=fluid_hip.mulberry32(7)) for s in ("passes", "fused")]
    for s in sims:
        s.multipleSplats(8); s.step(0.016666, 6)
    print(size, iters, {k: bool(np.array_equal(sims[0].read(k), sims[1].read(k))) for k in ("velocity", "pressure", "divergence", "curl", "dye")},
          "launches in a call of 1:", sims[1].schedule_info(1)["launches"])
    for s in sims: s.close()
P
echo "== A/B at 4096^2 / 50 =="
timeout 900 python tools/ab_env.py --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee $OUT/jacobi_chain_ab.txt
echo "== 8192^2 =="
timeout 600 python tools/ab_env.py --rounds 2 --args "--size 8192 --steps 40 --warmup 10" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1" 2>&1 | tee $OUT/jacobi_chain_ab_8192.txt
