"""TEST INFRASTRUCTURE — the display compositor at the sizes the reference SHIPS (script.js:59-85: sim 128, dye 1024, capture 512, bloom 256 with
8 iterations, sunrays 196; render(target) script.js:1296-1419): two fixtures rendered by the UNMODIFIED page under headless Chromium +
SwiftShader, a square canvas and a 2:1 one.  The 1024^2 dye the frame is rendered from is NOT stored (16 MB): the scenario — seeded random
splats, steps at power-of-two grid sizes — is bit-reproducible (tests/test_hip_vs_golden.py), so the tests replay it and hold the result to
every 8th row / column of the reference's dye and to the SHA-256 of all of it.  Stored: the 8-bit frame (all of it), the float frame and the
bloom buffer at every 2nd pixel, the sunrays buffer, and every 8th row / column of its mask (a dye-sized buffer).  Run in the build container only (needs kaleido + /root/reference)."""
import hashlib
import json
import os

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def save(name, scenario):
    res = live.run(scenario)
    fields = live.native_channels(res["fields"])
    dye = np.ascontiguousarray(fields["dye"], np.float32)
    payload = {"scenario": np.array(json.dumps(scenario)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]), "canvas": np.array(res["canvas"]),
               "frame8": res["frame8"], "frame_shape": np.array(res["frame"].shape), "frame_s2": res["frame"][::2, ::2].copy(),
               "bloom_shape": np.array(res["bloom"].shape), "bloom_s2": res["bloom"][::2, ::2].copy(), "sunrays": res["sunrays"], "mask_shape": np.array(res["mask"].shape), "mask_s8": res["mask"][::8, ::8].copy(),
               "bloom_levels": np.array(res["bloomLevels"]), "dye_s8": dye[::8, ::8].copy(), "dye_sha256": np.array(hashlib.sha256(dye.tobytes()).hexdigest()),
               "dye_max": np.array(float(np.abs(dye).max()))}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print("%-30s canvas %s dye %s frame %s bloom %s (%d levels) sunrays %s  max frame %.3f  file %.2f MB" % (
        name, res["canvas"], res["dye"], res["frame"].shape[:2], res["bloom"].shape[:2], len(res["bloomLevels"]), res["sunrays"].shape,
        res["frame"][..., :3].max(), os.path.getsize(os.path.join(OUT, name + ".npz")) / 1e6))


def main():
    dith = {"w": 8, "h": 8, "seed": 5}   # (the page's blue-noise PNG is an asset nobody ships here: a seeded 8 x 8 pattern on both sides, as in make_golden_display.py)
    cfg = {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 1024}   # the page's defaults; the display settings stay the page's own (nothing overridden)
    save("displayfull_default_512", {"canvasW": 1024, "canvasH": 1024, "config": cfg, "seed": 11, "randomSplats": 7, "steps": 4, "render": {"config": {}, "dither": dith}})
    # 2 : 1, so that every grid of the run is a power of two again (sim 256 x 128, dye 2048 x 1024: bit-reproducible; a 16 : 9 canvas gives a
    # 1820-wide dye whose state after three steps already carries 5e-5 of the rasteriser's coordinate jitter — tests/tolerances.py)
    save("displayfull_wide_1024x512", {"canvasW": 2048, "canvasH": 1024, "config": cfg, "seed": 12, "randomSplats": 6, "steps": 3, "render": {"config": {}, "dither": dith}})


if __name__ == "__main__":
    main()
