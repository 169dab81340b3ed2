#!/usr/bin/env python3
"""GPU box (lab): K6 as one more block of the chained pressure launch (FLUID_CHAIN_GS=1, libfluid_hip_probes.so) against the product library —
the five fields after a few steps, hashed, at shapes that take the chained launch (a width that is not a multiple of 4 among them).
Usage: python tools/chain_gs_check.py"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "webgl-fluid-simulation_amd")
CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import fluid_hip
W, H, IT, N = %d, %d, %d, %d
cfg = {"SIM_RESOLUTION": min(W, H), "DYE_RESOLUTION": min(W, H), "PRESSURE_ITERATIONS": IT}
with fluid_hip.FluidSim(canvas=(W, H), config=cfg, schedule="fused", random=fluid_hip.mulberry32(1234)) as sim:
    sim.multipleSplats(12)
    sim.step(0.016666, N)
    sim.step(0.016666, 1)
    si = sim.schedule_info(1)
    print(json.dumps({"chained": si.get("jacobi_chained"), "sim": list(sim.sim_size) if hasattr(sim, "sim_size") else None,
                      "h": {k: hashlib.sha256(sim.read(k).tobytes()).hexdigest()[:16] for k in ("velocity", "pressure", "divergence", "curl", "dye")}}))
"""


def run(env, W, H, it, n):
    r = subprocess.run([sys.executable, "-c", CHILD % (PKG, W, H, it, n)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    probes = os.path.join(PKG, "libfluid_hip_probes.so")
    bad = 0
    for (W, H, it, n) in [(4096, 4096, 50, 3), (4096, 4096, 20, 2), (4200, 3000, 50, 2), (3075, 3333, 30, 2), (3800, 2600, 45, 2), (3072, 3072, 200, 1)]:
        a = run({}, W, H, it, n)
        b = run({"FLUID_HIP_LIB": probes, "FLUID_CHAIN_GS": "1"}, W, H, it, n)
        c = run({"FLUID_HIP_LIB": probes, "FLUID_CHAIN_GS": "0"}, W, H, it, n)
        ok = "error" not in a and a.get("h") == b.get("h") == c.get("h")
        bad += 0 if ok else 1
        print("%5d x %5d, %3d iterations, %d + 1 steps: %s  chained %s  %s" % (W, H, it, n, "same bits" if ok else "MISMATCH", a.get("chained"), "" if ok else (a, b, c)), flush=True)
    print("chain_gs_check: %s" % ("OK" if not bad else "%d FAILED" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
