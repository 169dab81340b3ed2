#!/usr/bin/env python3
"""GPU box: per-pass device times and throughput of one configuration (sim / dye resolution, iterations) under different knob settings of the
lab build, one child process per setting, interleaved `--rounds` times.  For the cases bench.py does not cover (dye grid != sim grid).
Usage: python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 "" "FLUID_ADVECT_WY=4" ..."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666


def child(a):
    import fluid_hip
    cfg = {"SIM_RESOLUTION": a["sim"], "DYE_RESOLUTION": a["dye"], "PRESSURE_ITERATIONS": a["iters"]}
    side = max(a["sim"], a["dye"])
    with fluid_hip.FluidSim(canvas=(side, side), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
        sim.multipleSplats(10)
        sim.step(DT, 300)
        sim.sync()
        t0 = time.perf_counter()
        sim.step(DT, a["steps"])
        sim.sync()
        thr = 1e6 * (time.perf_counter() - t0) / a["steps"]
        sim.set_timing(True)
        sim.step(DT, 50)
        sim.sync()
        tm = sim.timings()
    print(json.dumps({"us_per_step": round(thr, 1), "pass_us": {k[:-3]: round(1e3 * v / max(tm["steps"], 1), 1) for k, v in tm.items() if k.endswith("_ms") and v}}))


def main():
    if os.environ.get("_ABP_CHILD"):
        return child(json.loads(os.environ["_ABP_CHILD"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--sim", type=int, default=1024)
    ap.add_argument("--dye", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    probes = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")
    print("# sim %d^2, dye %d^2, %d Jacobi iterations: us per step (back to back) and per-pass device times (timing mode)" % (a.sim, a.dye, a.iters), flush=True)
    for _ in range(a.rounds):
        for st in a.settings:
            env = dict(os.environ, _ABP_CHILD=json.dumps({"sim": a.sim, "dye": a.dye, "iters": a.iters, "steps": a.steps}))
            for kv in st.split():
                k, _, v = kv.partition("=")
                env[k] = v
            if st.split() and "FLUID_HIP_LIB" not in env:
                env["FLUID_HIP_LIB"] = probes
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print("[%-46s] %s" % (st, lines[-1] if lines else "FAILED " + r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
