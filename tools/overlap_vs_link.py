#!/usr/bin/env python3
"""GPU box (ONE GPU): does the interior-first overlap of the stripe / tile driver hide a link that TAKES TIME?  (VERDICT r03 item 4:
tests/fake_rccl copies instantly and in-process copies are intra-device, so the overlap had never met a slow link.)

ONE rank of a larger world runs alone (FAKE_RCCL_LOOPBACK=1: tests/fake_rccl serves every receive from this rank's own send to that
neighbour, so nothing waits for another rank and nothing else runs on the device): a MIDDLE stripe of three, or the centre tile of 3 x 3,
stepping through fluid_step_n — the native plan with grouped ncclSend / ncclRecv on the comm stream, csrc/fluid_stripes.cpp.
FAKE_RCCL_DELAY_US / FAKE_RCCL_GBPS make every exchange take latency + bytes / bandwidth on the comm stream (one spinning single-thread
kernel per group; the host never sleeps).  The rank then sees exactly a middle rank's timing: its own compute plus exchanges that take time.
What the link adds to the step is what the overlap does not hide; with overlap switched off (fluid_set_overlap 0) it must show in full —
that row is the probe's own check.

(The first form of this probe — several rank THREADS sharing the GPU, profiles/r04/overlap_vs_link_latency_visit2_raw.txt — measured
the device's arbitration between the ranks' queues instead: with 4 hardware queues two ranks' streams share queues, with 16 the ranks'
kernels thrash each other, 1.0 -> 1.5 ms per step with an instantaneous link.)

One child process per setting (the stand-in reads its environment once).
Usage: python tools/overlap_vs_link.py [--config stripe|tile|deep] > profiles/r04/overlap_vs_link_latency.txt"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666

CONFIGS = {
    # name: (canvas, SIM = DYE resolution, iterations, world, tiles_x, rank, halo, steps)
    "stripe": ((4096, 12288), 4096, 50, 3, 1, 1, 56, 60),       # weak scaling of configs[2]: the middle one of three 4096^2 stripes, 2 exchanges / step
    "tile": ((12288, 12288), 12288, 50, 9, 3, 4, 56, 60),       # configs[3]'s per-rank shape: the centre 4096^2 tile of 3 x 3 (eight neighbours)
    "deep": ((8192, 6144), 6144, 200, 3, 1, 1, 56, 16),         # configs[4]'s regime: 8192 x 2048 per rank, 200 iterations -> 5 exchanges / step
    "deep16": ((16384, 6144), 6144, 200, 3, 1, 1, 56, 12),      # configs[4]'s own per-rank shape: 16384 x 2048, 200 iterations
}


def child(a):
    import fluid_hip
    from fluid_hip import _abi
    from fluid_hip.sim import getResolution
    from fluid_hip.stripes import HipStripeEngine, new_comm_id
    canvas, res, iters, world, tx, r, halo, steps = CONFIGS[a["config"]]
    ty = world // tx
    cfg = dict(fluid_hip.DEFAULT_CONFIG, SIM_RESOLUTION=res, DYE_RESOLUTION=res, PRESSURE_ITERATIONS=iters)
    sim = getResolution(res, *canvas)
    rnd = fluid_hip.mulberry32(1234)
    splats = []
    for _ in range(20):
        c = fluid_hip.HSVtoRGB(rnd(), 1.0, 1.0)
        splats.append((rnd(), rnd(), 1000.0 * (rnd() - 0.5), 1000.0 * (rnd() - 0.5), c["r"] * 10.0, c["g"] * 10.0, c["b"] * 10.0))
    aspect = canvas[0] / canvas[1]
    radius = cfg["SPLAT_RADIUS"] / 100.0 * (aspect if aspect > 1 else 1.0)
    e = HipStripeEngine((sim["width"], sim["height"]), (sim["width"], sim["height"]), r // tx, ty, halo, _abi.SCHED_FUSED, 0, part_x=r % tx, parts_x=tx)
    e.use_own_stream()
    e.set_overlap(a["overlap"])
    e.comm_init(new_comm_id())      # loopback: returns at once
    if a.get("calibrate"):          # tests/test_stripes_gpu.py: does the start-up probe find the link the stand-in was given?
        lat, bw = e.calibrate_link(int(a.get("reps", 20)))
        e.close()
        print(json.dumps({"ok": True, "latency_us": lat, "GBps": bw}))
        return
    for s in splats:
        e.splat(*s, aspect, radius)
    e.step_n(40, DT, cfg)            # past the shader-clock dip of a load step (profiles/r04/first_steps.txt)
    e.sync()
    n0 = e.exchange_count()
    t0 = time.perf_counter()
    e.step_n(steps, DT, cfg)
    e.sync()
    el = time.perf_counter() - t0
    ex = (e.exchange_count() - n0) / steps
    fi = e.info("velocity")
    e.close()
    print(json.dumps({"ok": True, "ms_per_step": round(1e3 * el / steps, 4), "exchanges_per_step": ex, "owned": [fi.cols, fi.rows]}))


def main():
    if os.environ.get("_OVL_CHILD"):
        return child(json.loads(os.environ["_OVL_CHILD"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", action="append", default=None)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--quick", action="store_true", help="overlap on only; links: none, 60 us, 20 us + 100 GB/s, 20 us + 40 GB/s")
    args = ap.parse_args()
    knobs = {k: v for k, v in os.environ.items() if k.startswith("FLUID_") and k not in ("FLUID_RCCL_LIB",)}
    if knobs:
        print("## environment: %s" % knobs, flush=True)
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    for name in (args.config or list(CONFIGS)):
        canvas, res, iters, world, tx, rk, halo, steps = CONFIGS[name]
        print("## %s: rank %d of %d (%s) of a %dx%d grid, alone on the GPU (loopback), %d Jacobi iterations, halo %d, %d timed steps; link = latency per exchange (+ bytes / bandwidth)"
              % (name, rk, world, "%dx%d tiles" % (world // tx, tx) if tx > 1 else "stripes", canvas[0], canvas[1], iters, halo, steps), flush=True)
        base = {}
        for rnd in range(args.rounds):
            for overlap in ((1,) if args.quick else (1, 0)):
                for delay, gbps in (((0, 0), (60, 0), (20, 100), (20, 40)) if args.quick else ((0, 0), (60, 0), (200, 0), (20, 100), (20, 40))):
                    env = dict(os.environ, FLUID_RCCL_LIB=lib, FAKE_RCCL_LOOPBACK="1", _OVL_CHILD=json.dumps({"config": name, "overlap": overlap}))
                    env.pop("FAKE_RCCL_DELAY_US", None); env.pop("FAKE_RCCL_GBPS", None)
                    if delay:
                        env["FAKE_RCCL_DELAY_US"] = str(delay)
                    if gbps:
                        env["FAKE_RCCL_GBPS"] = str(gbps)
                    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
                    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    d = json.loads(lines[-1]) if lines else {"ok": False, "errors": [r.stderr[-300:]]}
                    label = "overlap %d  link %3d us%s" % (overlap, delay, " + bytes / %d GB/s" % gbps if gbps else "")
                    if not d.get("ok"):
                        print("   [%-40s] FAILED %s" % (label, d.get("errors")), flush=True)
                        continue
                    key = overlap
                    if delay == 0 and gbps == 0:
                        base[key] = d["ms_per_step"]
                    rel = 100.0 * (d["ms_per_step"] / base[key] - 1.0) if key in base else 0.0
                    print("   [%-40s] %.4f ms/step  (%+.1f %% against the instantaneous link of this round)  %.1f exchanges/step"
                          % (label, d["ms_per_step"], rel, d["exchanges_per_step"]), flush=True)


if __name__ == "__main__":
    main()
