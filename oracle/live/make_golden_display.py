"""TEST INFRASTRUCTURE — golden fixtures for the display compositor: the UNMODIFIED reference's
captureScreenshot() path — render(target) with bloom, sunrays, shading and the back-colour blend into a float FBO,
framebufferToTexture, normalizeTexture — on a state produced by its own splat()/step().  Saved per fixture: the dye
the frame was rendered from, the float frame, the 8-bit frame, the bloom and sunrays buffers, the sunrays mask.
Run in the build container only (needs kaleido + /root/reference)."""
import json
import os

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def save(name, scenario):
    res = live.run(scenario)
    fields = live.native_channels(res["fields"])
    payload = {"scenario": np.array(json.dumps(scenario)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]),
               "canvas": np.array(res["canvas"]), "frame": res["frame"], "frame8": res["frame8"], "bloom": res["bloom"],
               "sunrays": res["sunrays"], "mask": res["mask"], "bloom_levels": np.array(res["bloomLevels"]),
               "in_dye": fields["dye"], "in_velocity": fields["velocity"]}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **payload)
    print("%-34s frame %s bloom %s (%d levels) sunrays %s  max frame %.3f  mean a %.3f" % (
        name, res["frame"].shape[:2], res["bloom"].shape[:2], len(res["bloomLevels"]), res["sunrays"].shape,
        res["frame"][..., :3].max(), res["frame"][..., 3].mean()))


def main():
    base = {"canvasW": 512, "canvasH": 512, "config": {"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 96}, "seed": 77, "randomSplats": 5, "steps": 3}
    dith = {"w": 8, "h": 8, "seed": 5}
    small = {"CAPTURE_RESOLUTION": 80, "BLOOM_RESOLUTION": 48, "SUNRAYS_RESOLUTION": 40}
    save("display_default_80", dict(base, render={"config": dict(small), "dither": dith}))
    save("display_plain_80", dict(base, render={"config": dict(small, SHADING=False, BLOOM=False, SUNRAYS=False)}))
    save("display_shading_only_80", dict(base, render={"config": dict(small, BLOOM=False, SUNRAYS=False)}))
    save("display_bloom_only_80", dict(base, render={"config": dict(small, SHADING=False, SUNRAYS=False, BLOOM_ITERATIONS=3,
                                                                     BLOOM_INTENSITY=1.3, BLOOM_THRESHOLD=0.4, BLOOM_SOFT_KNEE=0.5), "dither": dith}))
    save("display_sunrays_only_80", dict(base, render={"config": dict(small, SHADING=False, BLOOM=False, SUNRAYS_WEIGHT=0.6)}))
    save("display_transparent_80", dict(base, render={"config": dict(small, TRANSPARENT=True), "dither": dith}))
    save("display_backcolor_wide_96x48",
         dict(base, canvasW=600, canvasH=300, render={"config": {"CAPTURE_RESOLUTION": 48, "BLOOM_RESOLUTION": 32, "SUNRAYS_RESOLUTION": 24,
                                                                  "BACK_COLOR": {"r": 40, "g": 90, "b": 200}}, "dither": dith}))


if __name__ == "__main__":
    main()
