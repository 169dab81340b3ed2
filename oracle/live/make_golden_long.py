"""TEST INFRASTRUCTURE — long-horizon fixtures from the live reference: 50 steps with CURL = 0 (the
trajectory stays comparable texel by texel) and 50 steps with CURL = 30 (trajectories decorrelate: only statistics are
comparable).  Run in the build container only."""
import json
import os

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def save(name, sc):
    res = live.run(sc)
    f = live.native_channels(res["fields"])
    payload = {"scenario": np.array(json.dumps(sc)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]), "canvas": np.array(res["canvas"]),
               "splats": np.array(res["splats"], dtype=np.float64).reshape(-1, 7), "draws": np.array(res["draws"])}
    for k, v in f.items():
        payload["out_" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    v = f["velocity"].astype(np.float64)
    print("%-24s kinetic %.6g  dye sum %.6g  max|v| %.5g" % (name, 0.5 * (v ** 2).sum(), f["dye"][..., :3].astype(np.float64).sum(), np.abs(v).max()))


def main():
    sq = {"canvasW": 512, "canvasH": 512}
    save("long50_curl0_64", dict(sq, config={"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "CURL": 0}, seed=31, randomSplats=4, steps=50))
    save("long50_curl30_64", dict(sq, config={"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}, seed=31, randomSplats=4, steps=50))


if __name__ == "__main__":
    main()
