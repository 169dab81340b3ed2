"""The input path: a recorded mouse / touch / keyboard stream replayed headless.
Golden `input_replay_600x300`: the same stream through the UNMODIFIED reference's own listeners
(script.js:1464-1530) + updateColors / applyInputs / step per frame (oracle/live/make_golden_inputs.py).
CPU: the JavaScript host (recording backend) and the Python host (recording subclass) must issue the reference's
splat() list EXACTLY (positions, deltas x SPLAT_FORCE, colours, Math.random call order).  GPU: final fields."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import scenario as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")


def golden():
    g, sc = S.load("input_replay_600x300")
    return g, sc, json.loads(str(g["frame_log"]))


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_js_host_replays_the_reference_event_stream_exactly():
    g, sc, frame_log = golden()
    args = {"canvas": {"width": int(g["canvas"][0]), "height": int(g["canvas"][1])}, "config": sc["config"], "seed": sc["seed"],
            "frames": sc["frames"]}
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node", "replay_host_logic.js"), json.dumps(args)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    splats = np.array([c[1:8] for c in out["calls"] if c[0] == "splat"], dtype=np.float64)
    assert splats.shape == g["splats"].shape
    assert np.array_equal(splats, g["splats"])                 # bit-for-bit the doubles the reference passed to splat()
    assert out["draws"] == int(g["draws"])
    assert out["frameLog"] == frame_log                        # splats per frame, PAUSED gate, pointer list growth, draw count
    steps = [c for c in out["calls"] if c[0] == "step"]
    assert len(steps) == sum(1 for f in frame_log if not f["paused"])   # paused frames apply inputs but do not step
    assert steps[2][1:3] == [1, 0.009]                         # the frame's own dt reaches step()


def test_python_host_replays_the_reference_event_stream_exactly():
    import fluid_hip
    from fluid_hip.sim import FluidSim
    g, sc, frame_log = golden()

    class Recording(FluidSim):   # host logic only: no device behind it
        def initFramebuffers(self):
            pass

        def splat(self, x, y, dx, dy, color):
            self.log.append([x, y, dx, dy, color["r"], color["g"], color["b"]])

        def step(self, dt, n=1):
            self.steps.append(dt)

        def close(self):
            pass

    draws = [0]
    rnd = fluid_hip.mulberry32(sc["seed"])

    def counted():
        draws[0] += 1
        return rnd()
    sim = Recording.__new__(Recording)     # host logic only: constructed without touching the device
    sim.log, sim.steps = [], []
    sim._lib = None
    sim._ctx = None
    sim.canvas = fluid_hip.Canvas(int(g["canvas"][0]), int(g["canvas"][1]))
    sim.config = dict(fluid_hip.DEFAULT_CONFIG, **sc["config"])
    sim.random, sim.splatStack, sim.pointers, sim.pixelRatio, sim._colorUpdateTimer = counted, [], [fluid_hip.sim.Pointer()], 1.0, 0.0
    log = []
    for f in sc["frames"]:
        for e in f.get("events", []):
            sim.dispatch(e)
        n0 = len(sim.log)
        sim.update(f["dt"])
        log.append({"splats": len(sim.log) - n0, "paused": bool(sim.config["PAUSED"]), "pointers": len(sim.pointers), "draws": draws[0]})
    assert np.array_equal(np.array(sim.log, dtype=np.float64), g["splats"])
    assert log == frame_log and draws[0] == int(g["draws"])


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["fused", "passes"])
def test_hip_replay_matches_reference_fields(schedule):
    import fluid_hip
    g, sc, _ = golden()
    with fluid_hip.FluidSim(canvas=(int(g["canvas"][0]), int(g["canvas"][1])), config=sc["config"], schedule=schedule,
                            random=fluid_hip.mulberry32(sc["seed"])) as sim:
        sim.replay(sc["frames"])
        got = sim.fields()
        assert sim.velocity.width == int(g["sim"][0]) and sim.dye.height == int(g["dye"][1])
    for k in S.FIELDS:
        want = g["out_" + k]
        # canvas 600 x 300 -> sim 64 x 32, dye 128 x 64: powers of two, so the live reference is bit-reproducible (tolerances.py)
        assert np.array_equal(got[k], want), (k, float(np.abs(got[k].astype(np.float64) - want).max()))
