#!/usr/bin/env python3
"""GPU box (ONE GPU): does the interior-first overlap of the stripe / tile driver hide a link that TAKES TIME?  (VERDICT r03 item 4:
tests/fake_rccl copies instantly and in-process copies are intra-device, so the overlap had never met a slow link.)

`world` rank threads step their stripe (or 2-D tile) contexts through fluid_step_n — the native plan with grouped ncclSend / ncclRecv on the
comm stream, csrc/fluid_stripes.cpp — against tests/fake_rccl with FAKE_RCCL_DELAY_US / FAKE_RCCL_GBPS: every receive then waits
latency + bytes / bandwidth on the receiver's comm stream (a spinning one-thread kernel; the host never sleeps).  All ranks share the one
GPU, so a step costs the SUM of the ranks' compute; what the link adds on top is what the overlap does not hide.  The ranks exchange at the
same points of their steps, so an exposed wait idles the whole device and shows in the wall time; with overlap switched off
(fluid_set_overlap 0) it must show in full — that row is the probe's own check.

One child process per setting (the stand-in reads its environment once).
Usage: python tools/overlap_vs_link.py [--config stripes2|tiles2x2|deep] > profiles/r04/overlap_vs_link_latency.txt"""
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
    # name: (canvas, SIM = DYE resolution, iterations, world, tiles_x, halo, steps)
    "stripes2": ((4096, 8192), 4096, 50, 2, 1, 56, 40),       # weak scaling of configs[2]: two stripes of 4096^2, 2 exchanges per step
    "tiles2x2": ((4096, 4096), 4096, 50, 4, 2, 56, 40),       # configs[3]'s shape at half the edge: 2 x 2 tiles of 2048^2, two-phase exchange
    "deep": ((8192, 4096), 4096, 200, 2, 1, 56, 12),          # configs[4]'s regime: 200 iterations -> 5 exchanges per step, two stripes of 8192 x 2048
}


def child(a):
    import fluid_hip
    from fluid_hip import _abi
    from fluid_hip.sim import getResolution
    from fluid_hip.stripes import HipStripeEngine, new_comm_id
    canvas, res, iters, world, tx, halo, steps = CONFIGS[a["config"]]
    ty = world // tx
    cfg = dict(fluid_hip.DEFAULT_CONFIG, SIM_RESOLUTION=res, DYE_RESOLUTION=res, PRESSURE_ITERATIONS=iters)
    sim = getResolution(res, *canvas)
    cid = new_comm_id()
    rnd = fluid_hip.mulberry32(1234)
    splats = []
    for _ in range(20):
        c = fluid_hip.HSVtoRGB(rnd(), 1.0, 1.0)
        splats.append((rnd(), rnd(), 1000.0 * (rnd() - 0.5), 1000.0 * (rnd() - 0.5), c["r"] * 10.0, c["g"] * 10.0, c["b"] * 10.0))
    aspect = canvas[0] / canvas[1]
    radius = cfg["SPLAT_RADIUS"] / 100.0 * (aspect if aspect > 1 else 1.0)
    bar = threading.Barrier(world)
    times, errs, exch = [0.0] * world, [], [0] * world

    def rank(r):
        try:
            e = HipStripeEngine((sim["width"], sim["height"]), (sim["width"], sim["height"]), r // tx, ty, halo, _abi.SCHED_FUSED, 0, part_x=r % tx, parts_x=tx)
            e.use_own_stream()
            e.set_overlap(a["overlap"])
            e.comm_init(cid)
            for s in splats:
                e.splat(*s, aspect, radius)
            e.step_n(5, DT, cfg)
            e.sync()
            n0 = e.exchange_count()
            bar.wait()
            t0 = time.perf_counter()
            e.step_n(steps, DT, cfg)
            e.sync()
            bar.wait()
            times[r] = time.perf_counter() - t0
            exch[r] = (e.exchange_count() - n0) / steps
            e.check_halo()
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(repr(ex))
            try:
                bar.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    print(json.dumps({"ok": not errs, "errors": errs, "ms_per_step": round(1e3 * max(times) / steps, 4), "exchanges_per_step": exch[0]}))


def main():
    if os.environ.get("_OVL_CHILD"):
        return child(json.loads(os.environ["_OVL_CHILD"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", action="append", default=None)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--hw-queues", type=int, default=16,
                    help="GPU_MAX_HW_QUEUES for the children.  The HIP runtime maps a process's streams onto 4 hardware queues by default; the "
                         "rank THREADS of this probe own 2-3 streams each, so with the default two ranks' compute and comm streams share queues "
                         "and wait for each other — an artefact of running several ranks in one process, which a one-rank-per-process run does "
                         "not have (own stream + comm stream + torch's: within 4).  0 = leave the default")
    args = ap.parse_args()
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    for name in (args.config or list(CONFIGS)):
        canvas, res, iters, world, tx, halo, steps = CONFIGS[name]
        print("## %s [GPU_MAX_HW_QUEUES=%s]: %dx%d global, %d ranks (%s) on ONE GPU, %d Jacobi iterations, halo %d, %d timed steps; link = latency per receive (+ bytes / bandwidth)"
              % (name, args.hw_queues or "default", canvas[0], canvas[1], world, "%dx%d tiles" % (world // tx, tx) if tx > 1 else "stripes", iters, halo, steps), flush=True)
        base = {}
        for rnd in range(args.rounds):
            for overlap in (1, 0):
                for delay, gbps in ((0, 0), (60, 0), (200, 0), (20, 100)):
                    env = dict(os.environ, FLUID_RCCL_LIB=lib, _OVL_CHILD=json.dumps({"config": name, "overlap": overlap}))
                    env.pop("FAKE_RCCL_DELAY_US", None); env.pop("FAKE_RCCL_GBPS", None)
                    if args.hw_queues:
                        env["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)
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
