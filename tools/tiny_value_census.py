#!/usr/bin/env python3
"""GPU box: how many 256 x 80 blocks of the bench workload's pressure / divergence fields hold a value that is tiny but not zero (0 < |x| <
2^-80) — the blocks a guarded fused-multiply-add form of the Jacobi update could NOT take (its single rounding differs from the
reference's two only when a result is subnormal; with every nonzero input >= 2^-80 no result of ten iterations can be).
Usage: python tools/tiny_value_census.py [size] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import torch
    import fluid_hip
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": iters}
    thr = 2.0 ** -80
    with fluid_hip.FluidSim(canvas=(N, N), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
        sim.multipleSplats(20)
        done = 0
        for n in (1, 4, 20, 25, 50, 150, 250, 500, 1000):
            sim.step(0.016666, n)
            sim.sync()
            done += n
            out = []
            for name in ("pressure", "divergence"):
                a = sim.device_view(name)[..., 0].abs()
                tiny = (a > 0) & (a < thr)
                H, W = tiny.shape
                hb, wb = H // 80 * 80, W // 256 * 256
                blocks = tiny[:hb, :wb].reshape(hb // 80, 80, wb // 256, 256).any(dim=3).any(dim=1)
                out.append("%s: %.4f %% of the texels tiny, %.1f %% of the 256x80 blocks hold one, %.1f %% of the texels exactly 0" % (
                    name, 100.0 * tiny.float().mean().item(), 100.0 * blocks.float().mean().item(), 100.0 * (a == 0).float().mean().item()))
            torch.cuda.synchronize()
            print("after %4d steps | %s | %s" % (done, out[0], out[1]), flush=True)


if __name__ == "__main__":
    main()
