#!/usr/bin/env python3
"""Time single pass groups of the fused schedule at several grid sizes (ns per texel): which kernels follow the memory system (time per
texel drops while the working set fits the 256 MB Infinity Cache) and which are bound inside the CU (time per texel constant).
Usage: tools/bench_pass.py [sizes ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def main():
    import fluid_hip
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192]
    for n in sizes:
        cfg = {"SIM_RESOLUTION": n, "DYE_RESOLUTION": n, "PRESSURE_ITERATIONS": 10}
        with fluid_hip.FluidSim(canvas=(n, n), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
            sim.multipleSplats(20)
            sim.step(0.016666, 3)
            out = {"size": n}
            for name, kw in (("advect", {}), ("gradsub", {}), ("curl_vorticity_divergence", {}), ("clear_jacobi", {"iters": 10})):
                reps = max(10, min(200, int(3e9 / (n * n * 48))))
                for _ in range(5):
                    sim.run_pass(name, **kw)
                sim.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    sim.run_pass(name, **kw)
                sim.sync()
                us = (time.perf_counter() - t0) / reps * 1e6
                out[name] = {"us": round(us, 1), "ns_per_texel": round(us * 1e3 / (n * n), 4)}
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
