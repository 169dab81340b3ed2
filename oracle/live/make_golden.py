"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/script.js) under Chromium/SwiftShader (oracle/live/live_reference.py).

Run in the build container only:  python oracle/live/make_golden.py
The fixtures travel to the GPU box; /root/reference does not.

Each .npz holds: `scenario` (JSON), optional `in_<field>` arrays that were injected as exact
fp32 state, `out_<field>` arrays read back with the reference's own framebufferToTexture
(native channel count), `sim`/`dye` sizes and the `splats` the reference itself issued.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import live_reference as live  # noqa: E402

OUT = os.path.normpath(os.path.join(HERE, "..", "..", "tests", "golden"))


def smooth(rng, h, w, nc, amp, k=4):
    """band-limited random field (sum of a few low modes) — like splat-generated data"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w, nc))
    for c in range(nc):
        for _ in range(k):
            fx, fy = rng.integers(1, 4, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            out[..., c] += rng.normal() * np.sin(2 * np.pi * fx * x / w + ph[0]) * np.cos(2 * np.pi * fy * y / h + ph[1])
    out *= amp / max(np.abs(out).max(), 1e-9)
    return out.astype(np.float32) if nc > 1 else out[..., 0].astype(np.float32)


def save(name, scenario, inject=None):
    sc = dict(scenario)
    if inject:
        sc["inject"] = inject
    res = live.run(sc)
    fields = live.native_channels(res["fields"])
    payload = {"scenario": np.array(json.dumps(scenario)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]),
               "canvas": np.array(res["canvas"]), "splats": np.array(res["splats"], dtype=np.float64).reshape(-1, 7),
               "draws": np.array(res["draws"])}
    for k, v in (inject or {}).items():
        payload["in_" + k] = v
    for k, v in fields.items():
        payload["out_" + k] = v
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print("%-28s sim %s dye %s  max|v| %.3g" % (name, res["sim"], res["dye"], np.abs(fields["velocity"]).max()))


def main():
    rng = np.random.default_rng(20240915)
    sq = {"canvasW": 512, "canvasH": 512}
    c64 = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}

    # --- driver-level scenarios -------------------------------------------------------
    save("splats_only_64", dict(sq, config=c64, seed=7, randomSplats=4, steps=0))
    save("step1_64", dict(sq, config=c64, seed=1234, randomSplats=3, steps=1))
    save("step5_curl0_64", dict(sq, config=dict(c64, CURL=0), seed=99, randomSplats=3, steps=5))
    save("step10_64", dict(sq, config=c64, seed=5, randomSplats=3, steps=10))
    save("step3_wide_64x32_dye96x48",
         dict(canvasW=800, canvasH=400, config={"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 48, "PRESSURE_ITERATIONS": 30},
              seed=3, randomSplats=2, splats=[[0.02, 0.97, 900.0, -850.0, 1.5, 0.2, 0.7]], steps=3))
    save("step3_tall_24x60",
         dict(canvasW=200, canvasH=500, config={"SIM_RESOLUTION": 24, "DYE_RESOLUTION": 24}, seed=11, randomSplats=3, steps=3))
    save("step5_sim32_dye128", dict(sq, config={"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 128}, seed=21, randomSplats=3, steps=5))
    save("step2_params_48",
         dict(sq, config={"SIM_RESOLUTION": 48, "DYE_RESOLUTION": 48, "CURL": 12.5, "PRESSURE": 0.55, "PRESSURE_ITERATIONS": 7,
                          "VELOCITY_DISSIPATION": 1.3, "DENSITY_DISSIPATION": 0.4, "SPLAT_RADIUS": 0.6},
              seed=8, randomSplats=2, steps=2, dt=0.011))
    save("splat_stream_20", dict(sq, config={"SIM_RESOLUTION": 16, "DYE_RESOLUTION": 16}, seed=1234, randomSplats=20, steps=0))
    save("resize_32_to_48_dye64",
         dict(sq, config={"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 32}, seed=2, randomSplats=3, steps=2,
              resizeTo={"SIM_RESOLUTION": 48, "DYE_RESOLUTION": 64}))

    # --- single passes on exactly injected fp32 state ---------------------------------
    def inj(h, w, kind):
        if kind == "noise":
            return {"velocity": rng.normal(0, 50, (h, w, 2)).astype(np.float32),
                    "pressure": rng.normal(0, 30, (h, w)).astype(np.float32),
                    "divergence": rng.normal(0, 30, (h, w)).astype(np.float32),
                    "curl": rng.normal(0, 30, (h, w)).astype(np.float32),
                    "dye": np.abs(rng.normal(0, 1, (h, w, 4))).astype(np.float32)}
        return {"velocity": smooth(rng, h, w, 2, 300.0), "pressure": smooth(rng, h, w, 1, 80.0),
                "divergence": smooth(rng, h, w, 1, 40.0), "curl": smooth(rng, h, w, 1, 60.0),
                "dye": np.abs(smooth(rng, h, w, 4, 2.0))}

    cfg = {"SIM_RESOLUTION": 40, "DYE_RESOLUTION": 40}
    for kind in ("smooth", "noise"):
        for p in ("curl", "vorticity", "divergence", "clear", "jacobi", "gradsub", "advect_velocity", "advect_dye"):
            save("pass_%s_%s_40" % (p, kind), dict(sq, config=cfg, passes=[p], steps=0), inject=inj(40, 40, kind))
    # non-square single passes (wall rules / aspect)
    cfgw = {"SIM_RESOLUTION": 24, "DYE_RESOLUTION": 24}
    for p in ("divergence", "jacobi", "advect_velocity", "advect_dye"):
        save("pass_%s_smooth_48x24" % p, dict(canvasW=600, canvasH=300, config=cfgw, passes=[p], steps=0), inject=inj(24, 48, "smooth"))
    # 5 Jacobi iterations in a row (bitwise gate for the temporally blocked kernel)
    save("pass_jacobi5_noise_40", dict(sq, config=cfg, passes=["jacobi"] * 5, steps=0), inject=inj(40, 40, "noise"))
    # the velocity clamp at +-1000 inside vorticity
    big = inj(40, 40, "smooth")
    big["velocity"] = (big["velocity"] * 3.6).astype(np.float32)
    save("pass_vorticity_clamp_40", dict(sq, config=cfg, passes=["vorticity"], steps=0), inject=big)


if __name__ == "__main__":
    main()
