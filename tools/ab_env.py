#!/usr/bin/env python3
"""GPU box: bench.py under different environment knobs, one child process per setting (the library reads its knobs once), interleaved
`--rounds` times so that box drift does not read as an effect.  Prints one line per run: steps/s, ms/step, per-pass ms.
Usage: python tools/ab_env.py [--rounds 2] [--args "--size 1024 --steps 2000 --warmup 200"] "K=V K2=V2" "K=V3" ...   ("" = no knob)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--args", default="--steps 100 --warmup 30")
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    for r in range(a.rounds):
        for st in a.settings:
            env = dict(os.environ)
            for kv in st.split():
                k, _, v = kv.partition("=")
                env[k] = v
            if st.split() and "FLUID_HIP_LIB" not in env:   # knobs are read by the lab build only (make PROBES=1); "" = the product library
                env["FLUID_HIP_LIB"] = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-budget", "0", "--no-traffic", "--no-steady"] + a.args.split()
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                ps = {k[:-3]: round(v * 1e3, 1) for k, v in d.get("pass_ms_per_step", {}).items() if v}
                rf = d.get("roofline", {})
                print("[%-28s] %9.1f steps/s  %.4f ms/step  jacobi launch %.1f us  passes(us) %s" % (
                    st, d["steps_per_sec"], d["ms_per_step"], rf.get("avg_launch_ms", 0) * 1e3, ps), flush=True)
            except Exception as ex:
                print("[%-28s] FAILED: %s %s" % (st, ex, p.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
