"""TEST INFRASTRUCTURE — fixtures for the fp16-STORAGE mode from the live reference.

On a real GPU the reference renders into half-float textures; the headless SwiftShader build keeps fp32.  The harness option
`halfTargets` emulates the former around the UNMODIFIED script.js: after every draw into a simulation framebuffer it rounds that
attachment to fp16 (nearest even) — the reference's own shaders and arithmetic, plus a 16F target's store rounding.  These are the
outputs the fp16 mode of the oracle (oracle.RefSim(storage="f16")) and of the HIP path (storage="f16") are held to.
Run in the build container only."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import live_reference as live  # noqa: E402
from make_golden import smooth  # noqa: E402

OUT = os.path.normpath(os.path.join(HERE, "..", "..", "tests", "golden"))


def half(a):
    with np.errstate(over="ignore"):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def save(name, scenario, inject=None):
    sc = dict(scenario, halfTargets=True)
    if inject:
        sc["inject"] = inject
    res = live.run(sc)
    fields = live.native_channels(res["fields"])
    payload = {"scenario": np.array(json.dumps(scenario)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]),
               "canvas": np.array(res["canvas"]), "splats": np.array(res["splats"], dtype=np.float64).reshape(-1, 7)}
    for k, v in (inject or {}).items():
        payload["in_" + k] = v
    for k, v in fields.items():
        assert np.array_equal(v, half(v)), (name, k)          # every stored value IS a half
        payload["out_" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print("%-34s sim %s dye %s  max|v| %.4g" % (name, res["sim"], res["dye"], np.abs(fields["velocity"]).max()))


def main():
    rng = np.random.default_rng(20260923)
    sq = {"canvasW": 512, "canvasH": 512}

    def inj(h, w, kind):   # inputs are halves already (an upload into a 16F texture rounds as well)
        if kind == "noise":
            d = {"velocity": rng.normal(0, 50, (h, w, 2)), "pressure": rng.normal(0, 30, (h, w)), "divergence": rng.normal(0, 30, (h, w)),
                 "curl": rng.normal(0, 30, (h, w)), "dye": np.abs(rng.normal(0, 1, (h, w, 4)))}
        else:
            d = {"velocity": smooth(rng, h, w, 2, 300.0), "pressure": smooth(rng, h, w, 1, 80.0), "divergence": smooth(rng, h, w, 1, 40.0),
                 "curl": smooth(rng, h, w, 1, 60.0), "dye": np.abs(smooth(rng, h, w, 4, 2.0))}
        return {k: half(v) for k, v in d.items()}

    cfg = {"SIM_RESOLUTION": 40, "DYE_RESOLUTION": 40}
    for kind in ("smooth", "noise"):
        for p in ("curl", "vorticity", "divergence", "clear", "jacobi", "gradsub", "advect_velocity", "advect_dye"):
            save("f16_pass_%s_%s_40" % (p, kind), dict(sq, config=cfg, passes=[p], steps=0), inject=inj(40, 40, kind))
    save("f16_pass_jacobi12_noise_40", dict(sq, config=cfg, passes=["jacobi"] * 12, steps=0), inject=inj(40, 40, "noise"))
    c64 = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}
    save("f16_splats_only_64", dict(sq, config=c64, seed=7, randomSplats=4, steps=0))
    save("f16_step1_64", dict(sq, config=c64, seed=1234, randomSplats=3, steps=1))
    save("f16_step3_curl0_64", dict(sq, config=dict(c64, CURL=0), seed=99, randomSplats=3, steps=3))
    save("f16_step3_64", dict(sq, config=c64, seed=5, randomSplats=3, steps=3))
    save("f16_step2_sim32_dye128", dict(sq, config={"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 128}, seed=21, randomSplats=3, steps=2))
    save("f16_step2_256_50", dict(sq, config={"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 50}, seed=1234, randomSplats=6, steps=2))


if __name__ == "__main__":
    main()
