#!/usr/bin/env python3
"""The reference's SHIPPING configuration on the HIP side (script.js:60-66: SIM_RESOLUTION 128, DYE_RESOLUTION 1024, 20 pressure
iterations) and its big sibling (sim 1024 / dye 4096): microseconds per step() through fluid_step — latency (one step, then a sync:
what an interactive host sees per frame) and throughput (steps back to back) — with the dye != sim fast advection kernels and, as the
A/B, with FLUID_ADVECT_FAST=0 (the general per-texel kernels of round 1/2).  Same bits either way (asserted on a field hash).
Usage: python tools/bench_shipping.py > profiles/r03/bench_shipping_defaults.json"""
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666
CASES = [("shipping defaults (script.js:60-66)", 128, 1024, 20), ("sim 1024 / dye 4096", 1024, 4096, 20)]


def child():
    import fluid_hip
    out = []
    for name, sim_res, dye_res, iters in CASES:
        cfg = {"SIM_RESOLUTION": sim_res, "DYE_RESOLUTION": dye_res, "PRESSURE_ITERATIONS": iters}
        with fluid_hip.FluidSim(canvas=(dye_res, dye_res), config=cfg, schedule="fused", random=fluid_hip.mulberry32(1234)) as sim:
            sim.multipleSplats(10)
            sim.step(DT, 3)
            h = hashlib.sha256()
            for k in ("velocity", "dye", "pressure"):
                h.update(sim.read(k).tobytes())
            sim.step(DT, 200)
            sim.sync()
            lat = []
            for _ in range(300):
                t0 = time.perf_counter()
                sim.step(DT, 1)
                sim.sync()
                lat.append(1e6 * (time.perf_counter() - t0))
            n = 2000 if sim_res <= 128 else 500
            sim.sync()
            t0 = time.perf_counter()
            sim.step(DT, n)
            sim.sync()
            thr = 1e6 * (time.perf_counter() - t0) / n
            sim.set_timing(True)
            sim.step(DT, 50)
            sim.sync()
            tm = sim.timings()
            sim.set_timing(False)
            out.append({"case": name, "sim": [sim_res, sim_res], "dye": [dye_res, dye_res], "iters": iters,
                        "latency_us_per_step_median": round(statistics.median(lat), 1), "latency_us_p10_p90": [round(sorted(lat)[30], 1), round(sorted(lat)[270], 1)],
                        "throughput_us_per_step": round(thr, 1), "fields_sha256_after_3_steps": h.hexdigest()[:16],
                        "pass_us_per_step": {k[:-3]: round(1e3 * v / max(tm["steps"], 1), 1) for k, v in tm.items() if k.endswith("_ms") and v}})
    print(json.dumps(out))


def main():
    if os.environ.get("_SHIP_CHILD"):
        return child()
    res = {}
    for label, env in (("fast", {}), ("general", {"FLUID_ADVECT_FAST": "0", "FLUID_HIP_LIB": os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")})):   # knobs: the lab build
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, _SHIP_CHILD="1", **env), capture_output=True, text=True)
        lines = [l for l in p.stdout.splitlines() if l.startswith("[")]
        res[label] = json.loads(lines[-1]) if lines else {"error": p.stderr[-400:]}
    same = all(a["fields_sha256_after_3_steps"] == b["fields_sha256_after_3_steps"] for a, b in zip(res["fast"], res["general"])) if isinstance(res["general"], list) and isinstance(res["fast"], list) else None
    print(json.dumps({"what": "us per step() of the reference's shipping configuration and its 8x sibling through fluid_step on one MI355X; "
                              "the reference itself: ~27 ms per step at the shipping defaults under SwiftShader (BASELINE.md section 2)",
                      "reference_ms_per_step_shipping_defaults": 27.0, "kernels_fast": res["fast"], "kernels_general_FLUID_ADVECT_FAST_0": res["general"],
                      "bitwise_equal_fast_vs_general": same}, indent=1))


if __name__ == "__main__":
    main()
