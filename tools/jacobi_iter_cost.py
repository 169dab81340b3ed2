#!/usr/bin/env python3
"""GPU box: the cost of ONE iteration inside a temporally blocked Jacobi launch, per tile shape and grid size: standalone launches of 1, 4,
7 and 10 iterations (fluid_pass_jacobi on random fields), microseconds each; slope = time per further iteration, intercept = the launch's
load / store / boundary part.  One child process per shape (FLUID_TB_VARIANT is read once).
Usage: python tools/jacobi_iter_cost.py "1024 4096" "0 8 16 18 19"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def child(N):
    import numpy as np
    import fluid_hip
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": 16, "PRESSURE_ITERATIONS": 10}
    rng = np.random.default_rng(1)
    out = {}
    with fluid_hip.FluidSim(canvas=(N, N), config=cfg, schedule="fused") as sim:
        sim.write("pressure", rng.normal(0, 30, (N, N)).astype(np.float32))
        sim.write("divergence", rng.normal(0, 30, (N, N)).astype(np.float32))
        for k in (1, 4, 7, 10):
            for _ in range(5):
                sim.run_pass("jacobi", iters=k)
            sim.sync()
            reps = 200 if N <= 2048 else 40
            t0 = time.perf_counter()
            for _ in range(reps):
                sim.run_pass("jacobi", iters=k)
            sim.sync()
            out[k] = round((time.perf_counter() - t0) / reps * 1e6, 2)
    print(json.dumps(out))


def main():
    if os.environ.get("_JIC_CHILD"):
        return child(int(sys.argv[1]))
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024 4096").split()]
    shapes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0 8 16 18 19").split()]
    for N in sizes:
        for v in shapes:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(N)], env=dict(os.environ, _JIC_CHILD="1", FLUID_TB_VARIANT=str(v), FLUID_HIP_LIB=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'webgl-fluid-simulation_amd', 'libfluid_hip_probes.so')),
                               capture_output=True, text=True)
            try:
                d = {int(k): x for k, x in json.loads(r.stdout.strip().splitlines()[-1]).items()}
                slope = (d[10] - d[1]) / 9.0
                print("N=%5d shape %2d: 1/4/7/10 iterations %6.2f %6.2f %6.2f %6.2f us  -> %.3f us per further iteration, %.2f us for the rest" % (
                    N, v, d[1], d[4], d[7], d[10], slope, d[1] - slope), flush=True)
            except Exception as ex:
                print("N=%d shape %d FAILED %s %s" % (N, v, ex, r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
