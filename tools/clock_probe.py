#!/usr/bin/env python3
"""GPU box: what shader clock does the chip run at while a given grid size is being stepped?  A child process steps the grid back to back
for a few seconds while this one samples `rocm-smi --showclocks` (sclk of device 0).  Small grids are chains of short, latency-bound
launches: if the power management clocks the chip down for them, per-iteration times in cycles and in microseconds diverge.
Usage: python tools/clock_probe.py 1024 2048 4096"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import fluid_hip
N = int(sys.argv[1])
cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": 50}
with fluid_hip.FluidSim(canvas=(N, N), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
    sim.multipleSplats(20)
    t_end = time.time() + float(sys.argv[2])
    n = 0
    k = max(20, int(4e9 / (N * N * 60)))
    t0 = time.time()
    while time.time() < t_end:
        sim.step(0.016666, k); sim.sync(); n += k
    print("steps/s %%.1f" %% (n / (time.time() - t0)))
"""


def sclk():
    r = subprocess.run(["rocm-smi", "--showclocks", "-d", "0"], capture_output=True, text=True)
    m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz", r.stdout, re.I)
    return int(m.group(1)) if m else None, r.stdout


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1024, 4096]
    v, raw = sclk()
    print("idle: sclk %s MHz" % v)
    if v is None:
        print(raw[:1500])
    for N in sizes:
        p = subprocess.Popen([sys.executable, "-c", CHILD % os.path.join(ROOT, "webgl-fluid-simulation_amd"), str(N), "8"], stdout=subprocess.PIPE, text=True)
        time.sleep(4.0)
        got = []
        for _ in range(6):
            got.append(sclk()[0])
            time.sleep(0.4)
        out, _ = p.communicate()
        print("%d^2: sclk samples %s MHz while stepping; %s" % (N, got, out.strip().splitlines()[-1] if out.strip() else ""))


if __name__ == "__main__":
    main()
