"""TEST INFRASTRUCTURE — what the reference's rasteriser hands its fragment shaders: the five varyings of the reference's own
baseVertexShader (script.js:440-459), rendered by a probe fragment shader into a float target inside the live page
(oracle_plotly.js `probeCoords`) at eleven grid sizes.  oracle/raster.py `varying()` restates the interpolation; tests/test_raster_mode.py
holds it to these dumps bit for bit.  Each varying depends on one pixel coordinate only (asserted here), so the dumps are 1-D.
Run in the build container only:  python oracle/live/make_golden_coords.py"""
import base64
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import live_reference as live  # noqa: E402

OUT = os.path.normpath(os.path.join(HERE, "..", "..", "tests", "golden"))
# (W, H, canvas W, canvas H): the canvas makes getResolution (script.js:1612-1624) produce that sim grid
SIZES = [(40, 40, 320, 320), (48, 24, 480, 240), (24, 60, 240, 600), (100, 50, 800, 400), (96, 48, 768, 384), (60, 60, 480, 480),
         (300, 150, 1200, 600), (250, 130, 1000, 520), (37, 53, 370, 530), (64, 64, 512, 512), (128, 64, 1024, 512)]


def main():
    payload = {"sizes": np.array([s[:2] for s in SIZES])}
    for W, H, cw, ch in SIZES:
        res = live.run({"canvasW": cw, "canvasH": ch, "config": {"SIM_RESOLUTION": min(W, H), "DYE_RESOLUTION": min(W, H)}, "steps": 0,
                        "probeCoords": True, "noDump": True})
        assert res["sim"] == [W, H], (res["sim"], W, H)
        a = np.frombuffer(base64.b64decode(res["coords"]["a"]), np.float32).reshape(H, W, 4)   # vUv.x, vUv.y, vL.x, vR.x
        b = np.frombuffer(base64.b64decode(res["coords"]["b"]), np.float32).reshape(H, W, 4)   # vT.y, vB.y, vL.y, vT.x
        for arr, axis in ((a[..., 0], 0), (a[..., 2], 0), (a[..., 3], 0), (a[..., 1], 1), (b[..., 0], 1), (b[..., 1], 1)):
            assert np.all(arr == (arr[0:1, :] if axis == 0 else arr[:, 0:1]))                  # a function of one pixel coordinate
        assert np.array_equal(b[..., 2], a[..., 1]) and np.array_equal(b[..., 3], a[..., 0])  # vL.y = vUv.y, vT.x = vUv.x
        key = "%dx%d_" % (W, H)
        payload.update({key + "uv_x": a[0, :, 0].copy(), key + "uv_y": a[:, 0, 1].copy(), key + "l_x": a[0, :, 2].copy(),
                        key + "r_x": a[0, :, 3].copy(), key + "t_y": b[:, 0, 0].copy(), key + "b_y": b[:, 0, 1].copy()})
        exact = np.array_equal(a[0, :, 0], (np.arange(W, dtype=np.float32) + np.float32(0.5)) / np.float32(W))
        print("%4d x %-4d vUv.x == (i + .5) / W: %s" % (W, H, exact))
    np.savez_compressed(os.path.join(OUT, "raster_varyings.npz"), **payload)


if __name__ == "__main__":
    main()
