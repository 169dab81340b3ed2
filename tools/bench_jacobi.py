#!/usr/bin/env python3
"""Tuning harness for the temporally blocked Jacobi kernel: for each tile-shape variant
(FLUID_TB_VARIANT, see fluid_kernels.hip) time `iters` Jacobi iterations at N^2 and check the result
bit-for-bit against the per-pass kernel.  Usage: tools/bench_jacobi.py [N] [iters]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def child(N, iters):
    import numpy as np
    import fluid_hip
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": 16, "PRESSURE_ITERATIONS": iters}
    rng = np.random.default_rng(1)
    p = rng.normal(0, 30, (N, N)).astype(np.float32)
    d = rng.normal(0, 30, (N, N)).astype(np.float32)
    out = {}
    ref = None
    for sched in ("passes", "fused"):
        with fluid_hip.FluidSim(canvas=(N, N), config=cfg, schedule=sched) as sim:
            sim.write("pressure", p); sim.write("divergence", d)
            sim.run_pass("jacobi", iters=iters)
            got = sim.read("pressure")
            if ref is None:
                ref = got
            else:
                out["bitwise_equal_to_passes"] = bool(np.array_equal(got, ref))
            for _ in range(3):
                sim.run_pass("jacobi", iters=iters)
            sim.sync()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                sim.run_pass("jacobi", iters=iters)
            sim.sync()
            out[sched + "_us"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
    print(json.dumps(out))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    if os.environ.get("_TB_CHILD"):
        return child(N, iters)
    names = ["8x10 h12/10", "8x8 h8", "8x11 h12/10", "8x12 h12/10", "8x12 h16/13", "8x16 h20/17", "16x12 h20/17", "4x24 h16/13"]
    only = [int(x) for x in os.environ.get("TB_VARIANTS", "").split()] or list(range(len(names)))
    for v, name in enumerate(names):
        if v not in only:
            continue
        env = dict(os.environ, FLUID_TB_VARIANT=str(v), FLUID_HIP_LIB=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'webgl-fluid-simulation_amd', 'libfluid_hip_probes.so'), _TB_CHILD="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(N), str(iters)], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        print("variant %d (%s) N=%d iters=%d: %s" % (v, name, N, iters, line), flush=True)


if __name__ == "__main__":
    main()
