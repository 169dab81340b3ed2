#!/usr/bin/env python3
"""Overhead of the stripe decomposition measured on ONE GPU: the same 4096 x 4096 grid as a single domain and as an
in-process stripe group (fluid_group_step_n: the native plan, ghost-row copies device-to-device) of 2 / 4 / 8
stripes, overlap on and off.  Same total work on the same device, so time ratio = redundant ghost rows + strip
launches + exchange bookkeeping (everything except the xGMI link itself).  Usage: tools/bench_group.py [N] [iters] [halo] [tall] [tiles_x]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    halo = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    tall = int(sys.argv[4]) if len(sys.argv) > 4 else 1     # grid = N x (N * tall): `tall` stripes of N x N each
    tiles_x = int(sys.argv[5]) if len(sys.argv) > 5 else 1  # > 1: grid = (N * tiles_x) x (N * tall / tiles_x), 2-D tiles of N x N
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": iters}
    steps, warm = 50, 5
    canvas = (N * tiles_x, N * tall // tiles_x)
    out = {"grid": list(canvas), "halo": halo, "tiles_x": tiles_x}
    cfg = dict(cfg, SIM_RESOLUTION=min(canvas), DYE_RESOLUTION=min(canvas))
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, random=fluid_hip.mulberry32(1234)) as one:
        one.multipleSplats(20)
        one.step(0.016666, warm); one.sync()
        t0 = time.perf_counter(); one.step(0.016666, steps); one.sync()
        out["single_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    for world in ((tall,) if tall > 1 else (2, 4, 8)):
        for overlap in (True, False):
            g = StripeGroup(world, canvas=canvas, config=cfg, halo=halo, random=fluid_hip.mulberry32(1234), overlap=overlap, tiles_x=tiles_x)
            try:
                g.multipleSplats(20)
                g.step(0.016666, warm); g.sync()
                t0 = time.perf_counter(); g.step(0.016666, steps); g.sync()
                ms = (time.perf_counter() - t0) / steps * 1e3
                g.check_halo()
                out["group%d_%s_ms" % (world, "overlap" if overlap else "sync")] = round(ms, 4)
                out["group%d_exchanges_per_step" % world] = g.engines[0].exchange_count() / (steps + warm)
                # device bytes one rank holds: every field array = (rows + 2 halo) x pitch texels; velocity, pressure, dye are double-buffered
                e, total = g.engines[0], 0
                for name, bufs in (("velocity", 2), ("pressure", 2), ("divergence", 1), ("curl", 1), ("dye", 2)):
                    fi = e.info(name)
                    total += bufs * (fi.rows + 2 * fi.halo) * fi.pitch * fi.channels * fi.bytes_per_channel
                out["group%d_MB_per_rank" % world] = round(total / 1e6, 1)
            finally:
                g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
