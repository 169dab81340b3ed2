#!/usr/bin/env python3
"""GPU box: how the step time of the bench workload develops with the step index — the driver's `--steps 20 --warmup 5` reads steps 6..25
after the splats, the steady figure steps ~300..2300.  Per window of steps: ms per step (wall, one sync per window) and the per-pass device
times (HIP events, library timing mode).  If only the advection changes, it is the flow (the back-traces of a fresh splat field are long and
incoherent); if every pass changes alike, it is the chip (clocks / power state).
Usage: python tools/step_timeline.py [size] [iters]"""
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
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": iters}
    with fluid_hip.FluidSim(canvas=(N, N), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
        sim.multipleSplats(20)
        sim.sync()
        done = 0
        for n in (5, 20, 25, 50, 100, 200, 400, 800, 1600):
            sim.set_timing(True)
            t0 = time.perf_counter()
            sim.step(DT, n)
            sim.sync()
            wall = 1e3 * (time.perf_counter() - t0) / n
            tm = sim.timings()
            sim.set_timing(False)
            ps = {k[:-3]: round(1e3 * v / n, 1) for k, v in tm.items() if k.endswith("_ms") and v}
            print("steps %5d..%5d  %.4f ms/step (wall, events on)  passes(us) %s" % (done + 1, done + n, wall, ps), flush=True)
            done += n
        # the same windows without the event marks
        for n in (20, 200, 2000):
            t0 = time.perf_counter()
            sim.step(DT, n)
            sim.sync()
            print("steps %5d..%5d  %.4f ms/step (wall)" % (done + 1, done + n, 1e3 * (time.perf_counter() - t0) / n), flush=True)
            done += n


if __name__ == "__main__":
    main()
