#!/usr/bin/env python3
"""GPU box: what the LAST partial round of tiles costs a temporally blocked Jacobi launch.  4096^2 is 1242 tiles of 8 waves x 10 rows on 512
workgroup slots: two full rounds and 218 tiles that run one per CU.  Timed here: ten iterations on 4096-wide grids of different heights
(tile counts around the multiples of 512), and the remaining rows alone with the smaller shapes — would a launch whose tail is made of
smaller tiles (two per CU) be shorter?  One child per (height, shape).  Usage: python tools/jacobi_tail_probe.py"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def child(W, H):
    import numpy as np
    import fluid_hip
    cfg = {"SIM_RESOLUTION": min(W, H), "DYE_RESOLUTION": 16, "PRESSURE_ITERATIONS": 10}
    rng = np.random.default_rng(1)
    with fluid_hip.FluidSim(canvas=(W, H), config=cfg, schedule="fused") as sim:
        fi = sim.read("pressure").shape
        sim.write("pressure", rng.normal(0, 30, fi).astype(np.float32))
        sim.write("divergence", rng.normal(0, 30, fi).astype(np.float32))
        out = {"shape": list(fi)}
        for k in (1, 10):
            for _ in range(5):
                sim.run_pass("jacobi", iters=k)
            sim.sync()
            reps = 60
            t0 = time.perf_counter()
            for _ in range(reps):
                sim.run_pass("jacobi", iters=k)
            sim.sync()
            out[k] = round((time.perf_counter() - t0) / reps * 1e6, 2)
    print(json.dumps(out))


def main():
    if os.environ.get("_JTP_CHILD"):
        return child(int(sys.argv[1]), int(sys.argv[2]))
    cases = [(H, 0) for H in (1690, 1750, 3370, 3430, 3490, 3730, 4096, 5050, 5110)] + [(726, v) for v in (0, 8, 9, 10)] + [(1446, v) for v in (0, 9, 10)]
    for H, v in cases:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "4096", str(H)], env=dict(os.environ, _JTP_CHILD="1", FLUID_TB_VARIANT=str(v), FLUID_HIP_LIB=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'webgl-fluid-simulation_amd', 'libfluid_hip_probes.so')),
                           capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("4096 x %4d shape %2d (field %s): 1 iteration %6.2f us, 10 iterations %6.2f us" % (H, v, d["shape"], d["1"], d["10"]), flush=True)
        except Exception as ex:
            print("H=%d shape %d FAILED %s %s" % (H, v, ex, r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
