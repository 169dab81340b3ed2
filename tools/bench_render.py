#!/usr/bin/env python3
"""GPU box: the display compositor (fluid_render = render(target), script.js:1296-1419) at the reference's SHIPPING sizes — sim 128 / dye 1024 on a
1024 x 1024 canvas, capture 512, bloom 256 x 8 iterations, sunrays 196, shading on (script.js:59-85) — a frame as the page composes it: one
step(dt), one render(target).  Device time per call (calls back to back, one sync), no readback.  Under `rocprofv3 --kernel-trace --stats` this
is the run whose per-kernel table is profiles/r06/render_kernel_stats.csv.
Usage: python tools/bench_render.py [n_frames]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    import fluid_hip
    out = {}
    for name, canvas, target in (("shipping: canvas 1024 x 1024, capture 512 x 512", (1024, 1024), (512, 512)),
                                 ("shipping, 2:1: canvas 2048 x 1024, capture 1024 x 512", (2048, 1024), (1024, 512))):
        with fluid_hip.FluidSim(canvas=canvas, random=fluid_hip.mulberry32(1234)) as sim:   # DEFAULT_CONFIG = the page's `config`
            sim.multipleSplats(10)
            sim.step(DT, 20)
            p = sim._display_params()
            for _ in range(20):
                sim._check(sim._lib.fluid_render(sim._ctx, target[0], target[1], C.byref(p)))
            sim.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                sim._check(sim._lib.fluid_render(sim._ctx, target[0], target[1], C.byref(p)))
            sim.sync()
            render_us = 1e6 * (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                sim.step(DT, 1)
                sim._check(sim._lib.fluid_render(sim._ctx, target[0], target[1], C.byref(p)))
            sim.sync()
            frame_us = 1e6 * (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                sim.step(DT, 1)
            sim.sync()
            step_us = 1e6 * (time.perf_counter() - t0) / n
            out[name] = {"render_us": round(render_us, 1), "step_us": round(step_us, 1), "step_plus_render_us": round(frame_us, 1), "calls": n,
                         "sim": [sim.sim_width, sim.sim_height] if hasattr(sim, "sim_width") else None, "target": list(target)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
