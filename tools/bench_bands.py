#!/usr/bin/env python3
"""Does the SAME grid run faster as several row bands stepped concurrently on one GPU?  Each of the eight launches of a step spends
about a tenth of its time filling and draining the chip; bands on separate streams (the in-process stripe group: own arrays with ghost
rows, the native plan, ghost rows by device-to-device copies) let one band's launch run into the tail of the other's.  Against that:
redundant ghost rows, the copies, twice the launches.  Usage: tools/bench_bands.py [N] [iters]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": iters}
    steps, warm = 200, 50
    out = {"grid": [N, N], "iters": iters}
    for rep in range(2):
        with fluid_hip.FluidSim(canvas=(N, N), config=cfg, random=fluid_hip.mulberry32(1234)) as one:
            one.multipleSplats(20)
            one.step(DT, warm); one.sync()
            t0 = time.perf_counter(); one.step(DT, steps); one.sync()
            out["single_ms_%d" % rep] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        for world, halo, overlap in ((2, 56, False), (2, 56, True), (2, 32, False), (4, 56, False)):
            g = StripeGroup(world, canvas=(N, N), config=cfg, halo=halo, random=fluid_hip.mulberry32(1234), overlap=overlap)
            try:
                g.multipleSplats(20)
                g.step(DT, warm); g.sync()
                t0 = time.perf_counter(); g.step(DT, steps); g.sync()
                ms = (time.perf_counter() - t0) / steps * 1e3
                g.check_halo()
                out["bands%d_halo%d_%s_ms_%d" % (world, halo, "overlap" if overlap else "sync", rep)] = round(ms, 4)
            finally:
                g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
