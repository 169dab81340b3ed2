#!/bin/bash
# round 5, visit 12: the chained Jacobi launch with tickets (order = start order) and the band-cyclic order (FLUID_CHAIN_BAND rows per band; 0 = first form)
OUT=$PWD/gpurun_out/r05v12; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== correctness: chain vs the per-pass schedule, every field, bands 4 / 2 / 0 =="
for band in 4 2 0; do
FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=$band timeout 300 python - <<'P' 2>&1 | grep -v amdgpu.ids | tail -4
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "webgl-fluid-simulation_amd"))
import numpy as np, fluid_hip
for size, iters in ((4096, 50), (4096, 47), (3200, 50), (6000, 30)):
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    sims = [fluid_hip.FluidSim(canvas=(size, size), config=cfg, schedule=s, random=fluid_hip.mulberry32(7)) for s in ("passes", "fused")]
    for s in sims:
        s.multipleSplats(8); s.step(0.016666, 12)
    print("band", os.environ["FLUID_CHAIN_BAND"], size, iters, all(bool(np.array_equal(sims[0].read(k), sims[1].read(k))) for k in ("velocity", "pressure", "divergence", "curl", "dye")))
    for s in sims: s.close()
P
done
echo "== A/B at 4096^2 / 50 (parity checked in every run) =="
timeout 900 python tools/ab_env.py --rounds 3 --args "--steps 100 --warmup 30 --no-profile-pass" "FLUID_SKIP_CURL=1" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=4" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=2" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=8" "FLUID_JACOBI_CHAIN=1 FLUID_CHAIN_BAND=1" 2>&1 | tee $OUT/jacobi_chain_bands.txt
echo "== probes (invalid results, no parity): no waits at all =="
timeout 300 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 30 --no-parity --no-profile-pass" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=4" "FLUID_JACOBI_CHAIN=4 FLUID_CHAIN_BAND=0" 2>&1 | tee -a $OUT/jacobi_chain_bands.txt
