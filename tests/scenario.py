"""Replays a tests/golden scenario (the JSON the live reference ran) on either implementation:
the CPU oracle (oracle.RefSim) or the HIP product (fluid_hip.FluidSim) through thin adapters, so
both are driven by exactly the same call sequence the reference executed."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("velocity", "pressure", "divergence", "curl", "dye")


def golden_names(prefix=""):
    """driver / single-pass scenarios (the input-replay and display fixtures have their own tests: test_input_replay.py, test_display.py, test_long_horizon.py)"""
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    return [n for n in names if not n.startswith(("input_", "display_", "displayfull_", "long50_", "big_", "f16_", "raster_"))]


def f16_golden_names():
    """fixtures of the fp16-storage mode: the live reference with half-float render targets emulated (oracle/live/make_golden_f16.py)"""
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "f16_*.npz")))


def load(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    sc = json.loads(str(g["scenario"]))
    return g, sc


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


class OracleAdapter:
    def __init__(self, O, canvas, config, seed, storage="f32"):
        self.O = O
        self.storage = storage
        self.sim = O.RefSim(canvas=canvas, config=config, seed=seed, storage=storage)

    def write(self, name, a):
        s = self.sim
        a = self.O.stored(np.ascontiguousarray(a, np.float32).copy(), self.storage)   # an upload into a half-float texture rounds too
        if name == "velocity": s.vel[0] = a
        elif name == "pressure": s.prs[0] = a
        elif name == "divergence": s.div = a
        elif name == "curl": s.curl = a
        elif name == "dye": s.dye[0] = a

    def multiple_splats(self, n):
        return self.sim.multiple_splats(n)

    def splat(self, x, y, dx, dy, rgb):
        self.sim.splat(x, y, dx, dy, rgb)

    def run_pass(self, p, dt):
        O, s = self.O, self.sim
        P = s.params()
        dt = O.f32(dt)
        st = lambda a: O.stored(a, self.storage)  # noqa: E731  (the pass output as the field keeps it)
        if p == "curl": s.curl = st(O.curl(s.vel[0]))
        elif p == "vorticity": s.vel[0] = st(O.vorticity(s.vel[0], s.curl, P.curl, dt))
        elif p == "divergence": s.div = st(O.divergence(s.vel[0]))
        elif p == "clear": s.prs[0] = st(O.clear(s.prs[0], P.pressure))
        elif p == "jacobi": s.prs[0] = st(O.jacobi(s.prs[0], s.div))
        elif p == "gradsub": s.vel[0] = st(O.gradsub(s.prs[0], s.vel[0]))
        elif p == "advect_velocity": s.vel[0] = st(O.advect(s.vel[0], s.vel[0], dt, P.velocity_dissipation))
        elif p == "advect_dye": s.dye[0] = st(O.advect(s.vel[0], s.dye[0], dt, P.density_dissipation))
        else: raise ValueError(p)

    def step(self, dt, n):
        self.sim.step(dt, n)

    def resize(self, cfg):
        self.sim.config.update(cfg)
        self.sim.init_framebuffers()

    def fields(self):
        return self.sim.fields()


class HipAdapter:
    def __init__(self, canvas, config, seed, schedule="fused", storage="f32"):
        import fluid_hip
        self.sim = fluid_hip.FluidSim(canvas=canvas, config=config, schedule=schedule, random=fluid_hip.mulberry32(seed), storage=storage)

    def write(self, name, a):
        self.sim.write(name, a)

    def multiple_splats(self, n):
        return self.sim.multipleSplats(n)

    def splat(self, x, y, dx, dy, rgb):
        self.sim.splat(x, y, dx, dy, {"r": rgb[0], "g": rgb[1], "b": rgb[2]})

    def run_pass(self, p, dt):
        self.sim.run_pass(p, dt=dt)

    def step(self, dt, n):
        self.sim.step(dt, n)

    def resize(self, cfg):
        self.sim.config.update(cfg)
        self.sim.initFramebuffers()

    def fields(self):
        return self.sim.fields()

    def close(self):
        self.sim.close()


def replay(adapter, g, sc):
    """same order as oracle/live/oracle_plotly.js: inject, random splats, listed splats, passes, steps, resize"""
    for k in FIELDS:
        if "in_" + k in g.files:
            adapter.write(k, g["in_" + k])
    log = []
    if sc.get("randomSplats"):
        log += adapter.multiple_splats(sc["randomSplats"])
    for s in sc.get("splats", []):
        adapter.splat(s[0], s[1], s[2], s[3], s[4:7])
        log.append(list(s))
    dt = sc.get("dt", 0.016666)
    for p in sc.get("passes", []):
        adapter.run_pass(p, dt)
    if sc.get("steps"):
        adapter.step(dt, sc["steps"])
    if sc.get("resizeTo"):
        adapter.resize(sc["resizeTo"])
    return adapter.fields(), np.array(log, dtype=np.float64).reshape(-1, 7)


def canvas_of(g):
    return (int(g["canvas"][0]), int(g["canvas"][1]))
