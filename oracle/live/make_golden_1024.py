"""TEST INFRASTRUCTURE — BASELINE.json configs[1] (1024^2 sim = dye, 50 Jacobi iterations) through the live reference:
2 steps after 6 seeded splats.  The full dump is 38 MB, so the fixture keeps every 8th row and column of each field
plus one full-resolution band of 16 rows — enough to hold an implementation to the reference texel by texel at this
size.  Run in the build container only."""
import json
import os

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def main():
    sc = {"canvasW": 1024, "canvasH": 1024, "config": {"SIM_RESOLUTION": 1024, "DYE_RESOLUTION": 1024, "PRESSURE_ITERATIONS": 50},
          "seed": 1234, "randomSplats": 6, "steps": 2}
    res = live.run(sc)
    f = live.native_channels(res["fields"])
    payload = {"scenario": np.array(json.dumps(sc)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]), "canvas": np.array(res["canvas"]),
               "splats": np.array(res["splats"], dtype=np.float64).reshape(-1, 7), "band": np.array([600, 616])}
    for k, v in f.items():
        payload["sub8_" + k] = np.ascontiguousarray(v[::8, ::8])
        payload["band_" + k] = np.ascontiguousarray(v[600:616])
        payload["absmax_" + k] = np.array(float(np.abs(v).max()))
    np.savez_compressed(os.path.join(OUT, "big_step2_1024.npz"), **payload)
    print("big_step2_1024: sim %s, max|v| %.4g, file %.1f KB" % (res["sim"], np.abs(f["velocity"]).max(),
                                                               os.path.getsize(os.path.join(OUT, "big_step2_1024.npz")) / 1024))


if __name__ == "__main__":
    main()
