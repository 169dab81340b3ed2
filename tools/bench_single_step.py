#!/usr/bin/env python3
"""GPU box: the per-frame path against the batched one.  A page calls step(dt) once per frame (script.js:1176-1186),
i.e. fluid_step; the benchmark calls fluid_step_n.  Since round 4 the launch that ends a call works ahead (the next call's curl / vorticity /
divergence, k_advect_cvd MODE 2), so repeated fluid_step should cost what fluid_step_n costs.  Per grid size: steps/s of
  one call for n steps | n calls of one step, back to back | the same n calls in the lab build with FLUID_RUN_AHEAD=0 (round 3's behaviour)
Usage: python tools/bench_single_step.py [sizes...]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666


def child(sizes):
    import fluid_hip
    out = {}
    for N in sizes:
        cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": 50}
        n = 4000 if N <= 1024 else 1500
        with fluid_hip.FluidSim(canvas=(N, N), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
            sim.multipleSplats(20)
            sim.step(DT, 300)
            sim.sync()
            t0 = time.perf_counter()
            sim.step(DT, n)
            sim.sync()
            batched = n / (time.perf_counter() - t0)
            for _ in range(300):
                sim.step(DT, 1)
            sim.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                sim.step(DT, 1)
            sim.sync()
            single = n / (time.perf_counter() - t0)
            info = sim.schedule_info(1)
        out[str(N)] = {"fluid_step_n": round(batched, 1), "fluid_step_x_n": round(single, 1), "ratio": round(single / batched, 4),
                       "runs_ahead": info["runs_ahead"], "launches_per_single_step": info["launches"]}
    print(json.dumps(out))


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]
    if os.environ.get("_BSS_CHILD"):
        return child(sizes)
    probes = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")
    for label, env in (("product library (works ahead)", {}), ("lab build, FLUID_RUN_AHEAD=0 (round 3's per-frame path)", {"FLUID_HIP_LIB": probes, "FLUID_RUN_AHEAD": "0"})):
        for rnd in range(2):
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(s) for s in sizes], env=dict(os.environ, _BSS_CHILD="1", **env), capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print("[%s] %s" % (label, lines[-1] if lines else "FAILED " + r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
