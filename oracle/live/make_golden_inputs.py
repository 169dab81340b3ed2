"""TEST INFRASTRUCTURE — golden fixture for the input path: a recorded pointer / touch /
keyboard event stream goes through the UNMODIFIED reference's own listeners (script.js:1464-1530), its
updateColors / applyInputs / step run per frame with prescribed dt, and the splat() calls it issued plus the
final fields are saved.  Run in the build container only (needs kaleido + /root/reference)."""
import json
import os

import numpy as np

import live_reference as live

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def frames():
    f = []
    ev = lambda **k: k   # noqa: E731
    # mouse: press, drag along a curve over 6 frames (one frame with two moves, one with none), release, move while up
    f.append({"dt": 0.016666, "events": [ev(type="mousedown", offsetX=120, offsetY=90)]})
    pts = [(150, 100), (190, 118), (240, 131), (300, 140), (366, 139)]
    for i, (x, y) in enumerate(pts):
        evs = [ev(type="mousemove", offsetX=x, offsetY=y)]
        if i == 2:
            evs.append(ev(type="mousemove", offsetX=x + 9, offsetY=y - 4))
        f.append({"dt": 0.016666 if i % 2 == 0 else 0.009, "events": evs})
    f.append({"dt": 0.016666, "events": []})
    f.append({"dt": 0.016666, "events": [ev(type="mouseup"), ev(type="mousemove", offsetX=400, offsetY=200)]})
    # touch: two fingers down, both move, one lifts, the other keeps moving
    t = lambda i, x, y: {"identifier": i, "pageX": x, "pageY": y}   # noqa: E731
    f.append({"dt": 0.012, "events": [ev(type="touchstart", touches=[t(7, 80, 250), t(9, 520, 60)])]})
    f.append({"dt": 0.016666, "events": [ev(type="touchmove", touches=[t(7, 101, 236), t(9, 498, 77)])]})
    f.append({"dt": 0.016666, "events": [ev(type="touchmove", touches=[t(7, 130, 225), t(9, 470, 99)])]})
    f.append({"dt": 0.016666, "events": [ev(type="touchend", touches=[t(9, 470, 99)]), ev(type="touchmove", touches=[t(7, 160, 210)])]})
    # keyboard: space queues random splats, P pauses one frame (inputs still applied), P resumes
    f.append({"dt": 0.016666, "events": [ev(type="keydown", code="Space", key=" ")]})
    f.append({"dt": 0.016666, "events": [ev(type="keydown", code="KeyP", key="p"), ev(type="mousedown", offsetX=300, offsetY=150)]})
    f.append({"dt": 0.016666, "events": [ev(type="mousemove", offsetX=310, offsetY=170)]})
    f.append({"dt": 0.016666, "events": [ev(type="keydown", code="KeyP", key="p"), ev(type="mousemove", offsetX=330, offsetY=180)]})
    return f


def main():
    sc = {"canvasW": 600, "canvasH": 300, "config": {"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 20,
                                                      "COLOR_UPDATE_SPEED": 10}, "seed": 2024, "frames": frames(), "steps": 0}
    res = live.run(sc)
    fields = live.native_channels(res["fields"])
    payload = {"scenario": np.array(json.dumps(sc)), "sim": np.array(res["sim"]), "dye": np.array(res["dye"]),
               "canvas": np.array(res["canvas"]), "splats": np.array(res["splats"], dtype=np.float64).reshape(-1, 7),
               "draws": np.array(res["draws"]), "frame_log": np.array(json.dumps(res["frameLog"]))}
    for k, v in fields.items():
        payload["out_" + k] = v
    np.savez_compressed(os.path.join(OUT, "input_replay_600x300.npz"), **payload)
    print("input_replay_600x300: canvas %s sim %s dye %s, %d splats, %d Math.random draws" % (
        res["canvas"], res["sim"], res["dye"], len(res["splats"]), res["draws"]))
    for i, fl in enumerate(res["frameLog"]):
        print("  frame %2d: %s" % (i, fl))


if __name__ == "__main__":
    main()
