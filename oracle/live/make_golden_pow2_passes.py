"""TEST INFRASTRUCTURE — single-pass fixtures at POWER-OF-TWO grid sizes, fp32 and fp16-storage, from the live reference.

At power-of-two sizes the reference's interpolated texel coordinates are exact (tests/tolerances.py), so every pass — the
LINEAR-fetching ones (advection) and the sqrt / divide ones (vorticity) included — is bit-reproducible: these fixtures are held with
array_equal, per pass, on smooth and white-noise inputs, square (64 x 64) and wide (128 x 64, aspect 2: the wall rules and the
advection's texel-size scaling differ per axis).  The 40 x 40 single-pass fixtures of make_golden.py / make_golden_f16.py stay as the
non-power-of-two cases with their jitter tolerances.
Run in the build container only:  python oracle/live/make_golden_pow2_passes.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G32  # noqa: E402
import make_golden_f16 as G16  # noqa: E402

PASSES = ("curl", "vorticity", "divergence", "clear", "jacobi", "gradsub", "advect_velocity", "advect_dye")


def state(rng, h, w, kind):
    if kind == "noise":
        return {"velocity": rng.normal(0, 50, (h, w, 2)).astype(np.float32), "pressure": rng.normal(0, 30, (h, w)).astype(np.float32),
                "divergence": rng.normal(0, 30, (h, w)).astype(np.float32), "curl": rng.normal(0, 30, (h, w)).astype(np.float32),
                "dye": np.abs(rng.normal(0, 1, (h, w, 4))).astype(np.float32)}
    return {"velocity": G32.smooth(rng, h, w, 2, 300.0), "pressure": G32.smooth(rng, h, w, 1, 80.0), "divergence": G32.smooth(rng, h, w, 1, 40.0),
            "curl": G32.smooth(rng, h, w, 1, 60.0), "dye": np.abs(G32.smooth(rng, h, w, 4, 2.0))}


def main():
    rng = np.random.default_rng(20260924)
    shapes = (("64", dict(canvasW=512, canvasH=512), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}, 64, 64),
              ("128x64", dict(canvasW=1024, canvasH=512), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}, 64, 128))
    for tag, canvas, cfg, h, w in shapes:
        for kind in ("smooth", "noise"):
            for p in PASSES:
                st = state(rng, h, w, kind)
                G32.save("pass_%s_%s_%s" % (p, kind, tag), dict(canvas, config=cfg, passes=[p], steps=0), inject=st)
                G16.save("f16_pass_%s_%s_%s" % (p, kind, tag), dict(canvas, config=cfg, passes=[p], steps=0),
                         inject={k: G16.half(v) for k, v in st.items()})


if __name__ == "__main__":
    main()
