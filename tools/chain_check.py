#!/usr/bin/env python3
"""GPU box: the chained pressure loop (k_jacobi_pchain / k_jacobi_tb_chain) against the one-kernel-per-pass schedule, bit for bit, under
lab-build knobs — one child process per setting (the library reads its knobs once).  Every field after 2 + 1 steps and the loop on its own.
Usage: python tools/chain_check.py [--shapes "4096x4096x50 4096x3072x47"] "K=V K2=V2" "K=V3" ...        ("" = the shipped defaults, product library)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = 0.016666


def child(shapes):
    sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
    import numpy as np
    import fluid_hip
    out = []
    for w, h, iters in shapes:
        cfg = {"SIM_RESOLUTION": min(w, h), "DYE_RESOLUTION": min(w, h), "PRESSURE_ITERATIONS": iters}
        sims = [fluid_hip.FluidSim(canvas=(w, h), config=cfg, schedule=s, random=fluid_hip.mulberry32(21)) for s in ("passes", "fused")]
        try:
            info = sims[1].schedule_info(3, DT)
            t0 = time.time()
            for s in sims:
                s.multipleSplats(5)
                s.step(DT, 2)
                s.multipleSplats(1)
                s.step(DT, 1)
            bad = [k for k in ("velocity", "pressure", "divergence", "curl", "dye") if not np.array_equal(sims[0].read(k), sims[1].read(k))]
            for s in sims:
                s.run_pass("jacobi", iters=iters)
            if not np.array_equal(sims[0].read("pressure"), sims[1].read("pressure")):
                bad.append("jacobi pass")
            out.append({"shape": [w, h, iters], "chained": bool(info["jacobi_chained"]), "equal": not bad, "differ": bad, "s": round(time.time() - t0, 1)})
        except Exception as ex:   # an error of the library (a chained launch that gave up) is a result too
            out.append({"shape": [w, h, iters], "error": str(ex)[:300]})
        finally:
            for s in sims:
                s.close()
    print(json.dumps(out))


def main():
    if os.environ.get("_CHAIN_CHECK_CHILD"):
        return child(json.loads(os.environ["_CHAIN_CHECK_CHILD"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096x50 4096x3072x47 4200x3000x11 3800x2600x80")
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    shapes = [[int(v) for v in s.split("x")] for s in a.shapes.split()]
    ok = True
    for st in a.settings:
        env = dict(os.environ)
        for kv in st.split():
            k, _, v = kv.partition("=")
            env[k] = v
        if st.split() and "FLUID_HIP_LIB" not in env:
            env["FLUID_HIP_LIB"] = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")
        env["_CHAIN_CHECK_CHILD"] = json.dumps(shapes)
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=1500)
        try:
            for r in json.loads(p.stdout.strip().splitlines()[-1]):
                good = r.get("equal", False)
                ok &= good
                print("[%-44s] %-16s %s" % (st, "x".join(map(str, r["shape"])), ("bitwise equal, chained %s, %.1f s" % (r["chained"], r["s"])) if good else ("MISMATCH " + json.dumps(r))), flush=True)
        except Exception as ex:
            ok = False
            print("[%-44s] FAILED: %s %s" % (st, ex, p.stderr[-400:]), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
