"""TEST INFRASTRUCTURE — single passes of the live reference at the HEADLINE size (4096^2), on exact fp32 state generated in the
page (`synth`, oracle_plotly.js; tests/synth.py regenerates it bit for bit):

  big_jacobi50_noise_4096      clearProgram + 50 x pressureProgram (script.js:508-519, 868-890, loop 1259-1266) on white-noise pressure
                               and divergence — bit-reproducible passes, so implementations are held to it with array_equal at the
                               size where the temporally blocked kernel has interior tiles, apron seams and all five launches;
  big_pass_<p>_<kind>_4096     ONE pass (curl, vorticity, divergence, gradsub, advect_velocity, advect_dye) on a 800-texel/s vortex,
                               `smooth` (no noise) and `noisy` (1 % white noise): the re-synchronised form of SURVEY Appendix C at the
                               width where the reference's LINEAR-fetch coordinate jitter is largest — splits the whole-step tolerance
                               at 4096^2 into "reference jitter" and "ours", pass by pass.

Kept per fixture: every 32nd row / column of the pass's output, three full-width bands (rows 2406-2413 straddle the tile seam of
the Jacobi kernel at row 2410; the bottom and top four rows are domain-edge tiles), max|field|.  Build container only."""
import json
import os
import sys

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
N = 4096
STRIDE, BANDS = 32, [[2406, 2414], [0, 4], [N - 4, N]]
OUTPUT_OF = {"curl": ["curl"], "vorticity": ["velocity"], "divergence": ["divergence"], "gradsub": ["velocity"],
             "advect_velocity": ["velocity"], "advect_dye": ["dye"], "jacobi": ["pressure"]}


def save(name, synth, passes, config, keep):
    sc = {"canvasW": N, "canvasH": N, "config": dict({"SIM_RESOLUTION": N, "DYE_RESOLUTION": N}, **config), "synth": synth,
          "passes": passes, "steps": 0, "sample": {"stride": STRIDE, "bands": BANDS}}
    res = live.run(sc, timeout=3600.0)
    payload = {"scenario": np.array(json.dumps(sc)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]),
               "bands": np.array(BANDS), "stride": np.array(STRIDE)}
    for k in keep:
        v = res["samples"][k]
        payload["sub_" + k] = v["sub"]
        payload["band_" + k] = v["band"]
        payload["absmax_" + k] = np.array(v["absmax"])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **payload)
    print("%-36s %s  file %.0f KB" % (name, {k: float(payload["absmax_" + k]) for k in keep}, os.path.getsize(path) / 1024), flush=True)


def main():
    only = [a for a in sys.argv[1:] if a != "steps"]
    if not only or "jacobi" in only:
        save("big_jacobi50_noise_4096",
             {"pressure": {"seed": 101, "noise": 60.0, "amp": [40.0], "R2": 0.3}, "divergence": {"seed": 102, "noise": 60.0, "amp": [25.0], "cx": 0.4, "R2": 0.2}},
             ["clear"] + ["jacobi"] * 50, {"PRESSURE": 0.8}, ["pressure"])
    vortex = {"amp": [10000.0, 4000.0], "cx": 0.47, "cy": 0.53, "R2": 0.08}   # peak |v| ~ 810 texel/s (a 13-texel back-trace); unequal amplitudes: the field is not divergence-free
    for kind, rel in (("smooth", 0.0), ("noisy", 0.01)):
        syn = {"velocity": dict(vortex, seed=201, noise=2 * 810.0 * rel),
               # scalar fields are channel 0 of the generator: a dipole -(y - cy) g, peak ~ 0.064 amp
               "curl": {"seed": 202, "noise": 2 * 64.0 * rel, "amp": [1000.0], "R2": 0.05},
               "pressure": {"seed": 203, "noise": 2 * 64.0 * rel, "amp": [1000.0], "cx": 0.52, "R2": 0.06},
               "dye": {"seed": 204, "noise": 2 * 1.0 * rel, "amp": [30.0, -20.0, 1.5, 0.9], "cx": 0.5, "cy": 0.5, "R2": 0.12}}
        for p in ("curl", "vorticity", "divergence", "gradsub", "advect_velocity", "advect_dye"):
            if only and p not in only:
                continue
            need = {"curl": ["velocity"], "vorticity": ["velocity", "curl"], "divergence": ["velocity"], "gradsub": ["velocity", "pressure"],
                    "advect_velocity": ["velocity"], "advect_dye": ["velocity", "dye"]}[p]
            save("big_pass_%s_%s_4096" % (p, kind), {k: syn[k] for k in need}, [p], {}, OUTPUT_OF[p])


def whole_steps():
    """big_step2_synth_4096: TWO whole step()s (script.js:1231-1294, CURL = 30, 50 iterations) from synthetic state, no splat — i.e.
    without the one operation whose result depends on the libm (exp, script.js:738).  At a power-of-two width every texel-centre
    coordinate is exact in fp32, so the reference's rasteriser-interpolated coordinates carry no jitter and the WHOLE STEP is
    bit-reproducible: every field of the restatement and of the HIP path is held to this fixture with array_equal."""
    rel = 0.01
    syn = {"velocity": {"amp": [10000.0, 4000.0], "cx": 0.47, "cy": 0.53, "R2": 0.08, "seed": 301, "noise": 2 * 810.0 * rel},
           "pressure": {"seed": 303, "noise": 2 * 64.0 * rel, "amp": [1000.0], "cx": 0.52, "R2": 0.06},
           "dye": {"seed": 304, "noise": 2 * 1.0 * rel, "amp": [30.0, -20.0, 1.5, 0.9], "cx": 0.5, "cy": 0.5, "R2": 0.12}}
    sc = {"canvasW": N, "canvasH": N, "config": {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": 50}, "synth": syn,
          "steps": 2, "sample": {"stride": STRIDE, "bands": BANDS}}
    res = live.run(sc, timeout=3600.0)
    payload = {"scenario": np.array(json.dumps(sc)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]),
               "bands": np.array(BANDS), "stride": np.array(STRIDE)}
    for k, v in res["samples"].items():
        payload["sub_" + k] = v["sub"]
        payload["band_" + k] = v["band"]
        payload["absmax_" + k] = np.array(v["absmax"])
    path = os.path.join(OUT, "big_step2_synth_4096.npz")
    np.savez_compressed(path, **payload)
    print("big_step2_synth_4096 %s file %.0f KB" % ({k: float(payload["absmax_" + k]) for k in res["samples"]}, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if "steps" in sys.argv[1:]:
        whole_steps()
    if sys.argv[1:] != ["steps"]:
        main()
