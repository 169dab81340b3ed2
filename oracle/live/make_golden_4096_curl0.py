"""TEST INFRASTRUCTURE — the headline grid over a longer horizon: 4096^2, 50 Jacobi iterations, CURL = 0 (no vorticity
confinement: the trajectory stays non-chaotic, so ten steps can still be compared texel by texel), 10 steps after 20 seeded splats.  A full dump would be 600 MB, so the harness samples
in the page: every 32nd row and column of each field, one full-resolution band of 8 rows, and max|field|.  That holds an
implementation to the reference texel by texel at the size the benchmark runs at.  Run in the build container only
(~2.3 s per step under SwiftShader, ~1.5 GB of browser memory)."""
import json
import os

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
STRIDE, BAND = 32, (2400, 2408)


def main():
    sc = {"canvasW": 4096, "canvasH": 4096, "config": {"SIM_RESOLUTION": 4096, "DYE_RESOLUTION": 4096, "PRESSURE_ITERATIONS": 50, "CURL": 0},
          "seed": 1234, "randomSplats": 20, "steps": 10, "sample": {"stride": STRIDE, "band": list(BAND)}}
    res = live.run(sc, timeout=3600.0)
    payload = {"scenario": np.array(json.dumps(sc)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]), "canvas": np.array(res["canvas"]),
               "splats": np.array(res["splats"], dtype=np.float64).reshape(-1, 7), "band": np.array(BAND), "stride": np.array(STRIDE)}
    for k, v in res["samples"].items():
        payload["sub_" + k] = v["sub"]
        payload["band_" + k] = v["band"]
        payload["absmax_" + k] = np.array(v["absmax"])
    path = os.path.join(OUT, "big_step10_curl0_4096.npz")
    np.savez_compressed(path, **payload)
    print("big_step10_curl0_4096: sim %s, max|v| %.4g, ms %s, file %.1f KB" % (res["sim"], payload["absmax_velocity"], res.get("ms"), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
