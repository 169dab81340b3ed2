#!/usr/bin/env python3
"""GPU box: the pressure loop ALONE (pass_jacobi through the per-pass entry point: what a step runs between its divergence and its gradient
subtract) at arbitrary W x H x iterations, under lab-build knobs — one child process per setting, interleaved `--rounds` times.  For the map of
where the chained launch pays: width (tiles per row, panels) against whether the loop's set (12 B/texel) fits the Infinity Cache.
Usage: python tools/bench_loop.py [--rounds 2] [--shapes "4096x4096x50 8192x2048x50"] "K=V" "K=V2" ...      ("" = the product library)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(shapes):
    sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
    import numpy as np
    import fluid_hip
    out = []
    for w, h, iters in shapes:
        cfg = {"SIM_RESOLUTION": min(w, h), "DYE_RESOLUTION": 16, "PRESSURE_ITERATIONS": iters}
        with fluid_hip.FluidSim(canvas=(w, h), config=cfg, schedule="fused") as sim:
            rng = np.random.default_rng(1)
            sim.write("pressure", rng.normal(0, 30, (h, w)).astype(np.float32))
            sim.write("divergence", rng.normal(0, 30, (h, w)).astype(np.float32))
            info = sim.schedule_info(1, 0.016666)
            for _ in range(5):
                sim.run_pass("jacobi", iters=iters)
            sim.sync()
            reps = max(10, int(2e9 / (w * h * iters)))
            t0 = time.perf_counter()
            for _ in range(reps):
                sim.run_pass("jacobi", iters=iters)
            sim.sync()
            us = (time.perf_counter() - t0) / reps * 1e6
            out.append({"shape": [w, h, iters], "us": round(us, 1), "ns_per_mtexel_iter": round(us * 1e3 / (w * h * iters / 1e6), 2), "chained": bool(info["jacobi_chained"])})
    print(json.dumps(out))


def main():
    if os.environ.get("_LOOP_CHILD"):
        return child(json.loads(os.environ["_LOOP_CHILD"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--shapes", default="4096x4096x50 8192x2048x50")
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    shapes = [[int(v) for v in s.split("x")] for s in a.shapes.split()]
    for _ in range(a.rounds):
        for st in a.settings:
            env = dict(os.environ)
            for kv in st.split():
                k, _, v = kv.partition("=")
                env[k] = v
            if st.split() and "FLUID_HIP_LIB" not in env:
                env["FLUID_HIP_LIB"] = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")
            env["_LOOP_CHILD"] = json.dumps(shapes)
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=1500)
            try:
                for r in json.loads(p.stdout.strip().splitlines()[-1]):
                    print("[%-36s] %-16s %9.1f us  %6.2f ns per Mtexel-iteration  (chained per schedule_info: %s)" % (st, "x".join(map(str, r["shape"])), r["us"], r["ns_per_mtexel_iter"], r["chained"]), flush=True)
            except Exception as ex:
                print("[%-36s] FAILED: %s %s" % (st, ex, p.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
