"""TEST INFRASTRUCTURE — Python face of the CPU oracle (oracle/fluid_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
It restates, independently of the product host code, the reference's host-side logic
around the passes: getResolution (script.js:1612-1624), initFramebuffers (982-1010),
resizeDoubleFBO (1116-1126), splat/correctRadius (1441-1462), multipleSplats (1427-1439),
generateColor/HSVtoRGB (1565-1595) and step (1231-1294).

Parity: pinned against the live reference through tests/golden (see fluid_oracle.c header).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfluid_oracle.so")


class Win(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("g0", C.c_int), ("rows", C.c_int)]


class Params(C.Structure):
    _fields_ = [("curl", C.c_float), ("pressure", C.c_float), ("iterations", C.c_int),
                ("velocity_dissipation", C.c_float), ("density_dissipation", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "fluid_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["bash", os.path.join(HERE, "build.sh")], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
FP = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        WP = C.POINTER(Win)
        L.fo_curl.argtypes = [WP, FP, FP, C.c_int, C.c_int]
        L.fo_vorticity.argtypes = [WP, FP, FP, FP, C.c_float, C.c_float, C.c_int, C.c_int]
        L.fo_divergence.argtypes = [WP, FP, FP, C.c_int, C.c_int]
        L.fo_clear.argtypes = [WP, FP, FP, C.c_float, C.c_int, C.c_int]
        L.fo_jacobi.argtypes = [WP, FP, FP, FP, C.c_int, C.c_int]
        L.fo_gradsub.argtypes = [WP, FP, FP, FP, C.c_int, C.c_int]
        L.fo_advect.argtypes = [WP, FP, WP, FP, C.c_int, FP, C.c_float, C.c_float, C.c_int, C.c_int]
        L.fo_advect.restype = C.c_long
        L.fo_splat.argtypes = [WP, FP, C.c_int, FP] + [C.c_float] * 7 + [C.c_int, C.c_int]
        L.fo_resample.argtypes = [WP, FP, C.c_int, WP, FP]
        L.fo_step.argtypes = [C.c_int] * 4 + [C.POINTER(FP)] * 3 + [FP, FP, C.c_float, C.POINTER(Params)]
        L.fo_step_f16.argtypes = L.fo_step.argtypes
        L.fo_round_half.argtypes = [FP, C.c_long]
        L.fo_num_threads.restype = C.c_int
        _lib = L
    return _lib


def round_half(a: np.ndarray) -> np.ndarray:
    """what a write to a half-float render target keeps of fp32 values: nearest fp16, ties to even (fo_round_half)"""
    out = np.ascontiguousarray(a, np.float32).copy()
    lib().fo_round_half(_p(out), out.size)
    return out


def stored(a: np.ndarray, storage: str) -> np.ndarray:
    """a pass output as the field keeps it: unchanged (storage "f32") or rounded to fp16 ("f16")"""
    if storage not in ("f32", "f16"):
        raise ValueError(storage)
    return round_half(a) if storage == "f16" else a


def _p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(FP)


def f32(x: float) -> float:
    """JS number -> GLSL float uniform (gl.uniform1f rounds the double to fp32)."""
    return float(np.float32(x))


# ---- per-pass wrappers; arrays are [rows, W(, nc)] windows, whole domain by default ----
def _win(W: int, H: int, g0: int = 0, rows: Optional[int] = None) -> Win:
    return Win(W, H, g0, H if rows is None else rows)


def _rng(arr, ra, rb):
    return (0 if ra is None else ra, arr.shape[0] if rb is None else rb)


def curl(vel, H=None, g0=0, ra=None, rb=None):
    rows, W = vel.shape[:2]
    w = _win(W, rows if H is None else H, g0, rows)
    out = np.zeros((rows, W), np.float32)
    lib().fo_curl(C.byref(w), _p(vel), _p(out), *_rng(vel, ra, rb))
    return out


def vorticity(vel, crl, curl_strength, dt, H=None, g0=0, ra=None, rb=None):
    rows, W = vel.shape[:2]
    w = _win(W, rows if H is None else H, g0, rows)
    out = vel.copy()
    lib().fo_vorticity(C.byref(w), _p(vel), _p(crl), _p(out), curl_strength, dt, *_rng(vel, ra, rb))
    return out


def divergence(vel, H=None, g0=0, ra=None, rb=None):
    rows, W = vel.shape[:2]
    w = _win(W, rows if H is None else H, g0, rows)
    out = np.zeros((rows, W), np.float32)
    lib().fo_divergence(C.byref(w), _p(vel), _p(out), *_rng(vel, ra, rb))
    return out


def clear(p, value, ra=None, rb=None):
    rows, W = p.shape
    w = _win(W, rows)
    out = p.copy()
    lib().fo_clear(C.byref(w), _p(p), _p(out), value, *_rng(p, ra, rb))
    return out


def jacobi(p, div, H=None, g0=0, ra=None, rb=None):
    rows, W = p.shape
    w = _win(W, rows if H is None else H, g0, rows)
    out = p.copy()
    lib().fo_jacobi(C.byref(w), _p(p), _p(div), _p(out), *_rng(p, ra, rb))
    return out


def gradsub(p, vel, H=None, g0=0, ra=None, rb=None):
    rows, W = p.shape
    w = _win(W, rows if H is None else H, g0, rows)
    out = vel.copy()
    lib().fo_gradsub(C.byref(w), _p(p), _p(vel), _p(out), *_rng(p, ra, rb))
    return out


def advect(vel, src, dt, dissipation, vH=None, vg0=0, sH=None, sg0=0, ra=None, rb=None, return_misses=False):
    vrows, vW = vel.shape[:2]
    srows, sW = src.shape[:2]
    nc = src.shape[2]
    vw = _win(vW, vrows if vH is None else vH, vg0, vrows)
    sw = _win(sW, srows if sH is None else sH, sg0, srows)
    out = src.copy()
    m = lib().fo_advect(C.byref(vw), _p(vel), C.byref(sw), _p(src), nc, _p(out), dt, dissipation, *_rng(src, ra, rb))
    return (out, m) if return_misses else out


def splat(base, x, y, aspect, radius, color, H=None, g0=0, ra=None, rb=None):
    rows, W, nc = base.shape
    w = _win(W, rows if H is None else H, g0, rows)
    out = base.copy()
    c = list(color) + [0.0] * (3 - len(color))
    lib().fo_splat(C.byref(w), _p(base), nc, _p(out), x, y, aspect, radius, c[0], c[1], c[2], *_rng(base, ra, rb))
    return out


def resample(src, newW, newH):
    H, W, nc = src.shape
    sw, dw = _win(W, H), _win(newW, newH)
    out = np.zeros((newH, newW, nc), np.float32)
    lib().fo_resample(C.byref(sw), _p(src), nc, C.byref(dw), _p(out))
    return out


# ---- host-side logic of the reference, restated ---------------------------------------
def mulberry32(seed: int):
    """The PRNG the live harness installs in place of Math.random (oracle_plotly.js)."""
    state = [seed & 0xFFFFFFFF]

    def imul(a, b):
        return ((a & 0xFFFFFFFF) * (b & 0xFFFFFFFF)) & 0xFFFFFFFF

    def rnd() -> float:
        state[0] = (state[0] + 0x6D2B79F5) & 0xFFFFFFFF
        s = state[0]
        t = imul(s ^ (s >> 15), 1 | s)
        t = ((t + imul(t ^ (t >> 7), 61 | t)) & 0xFFFFFFFF) ^ t
        return ((t ^ (t >> 14)) & 0xFFFFFFFF) / 4294967296.0

    return rnd


def hsv_to_rgb(h: float, s: float, v: float) -> Tuple[float, float, float]:
    """script.js:1573-1595"""
    i = math.floor(h * 6)
    f = h * 6 - i
    p = v * (1 - s)
    q = v * (1 - f * s)
    t = v * (1 - (1 - f) * s)
    return [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)][int(i % 6)]


def js_round(x: float) -> int:
    """Math.round: half away from -inf, i.e. floor(x + 0.5)."""
    return int(math.floor(x + 0.5))


def get_resolution(resolution: float, canvas_w: int, canvas_h: int) -> Tuple[int, int]:
    """script.js:1612-1624 -> (width, height)"""
    aspect = canvas_w / canvas_h
    if aspect < 1:
        aspect = 1.0 / aspect
    mn = js_round(resolution)
    mx = js_round(resolution * aspect)
    return (mx, mn) if canvas_w > canvas_h else (mn, mx)


DEFAULT_CONFIG = dict(SIM_RESOLUTION=128, DYE_RESOLUTION=1024, DENSITY_DISSIPATION=1.0, VELOCITY_DISSIPATION=0.2,
                      PRESSURE=0.8, PRESSURE_ITERATIONS=20, CURL=30, SPLAT_RADIUS=0.25, SPLAT_FORCE=6000)


class RefSim:
    """Whole-domain CPU simulation with the reference's driver semantics."""

    def __init__(self, canvas: Tuple[int, int] = (512, 512), config: Optional[dict] = None, seed: int = 1234, storage: str = "f32"):
        """storage "f16": every pass output is rounded to fp16 as it is stored (the reference's half-float textures)"""
        if storage not in ("f32", "f16"):
            raise ValueError(storage)
        self.storage = storage
        self.canvas = canvas
        self.config = dict(DEFAULT_CONFIG)
        self.config.update(config or {})
        self.random = mulberry32(seed)
        self.vel = self.dye = None
        self.init_framebuffers()

    # script.js:982-1010 (+1116-1126): dye/velocity preserved through a bilinear copy, the rest zeroed
    def init_framebuffers(self):
        sw, sh = get_resolution(self.config["SIM_RESOLUTION"], *self.canvas)
        dw, dh = get_resolution(self.config["DYE_RESOLUTION"], *self.canvas)
        if self.dye is None:
            d = np.zeros((dh, dw, 4), np.float32)
            d[..., 3] = 1.0  # clear colour alpha (script.js:136, 1059)
            self.dye = [d, d.copy()]
        elif self.dye[0].shape[:2] != (dh, dw):
            d2 = np.zeros((dh, dw, 4), np.float32)
            d2[..., 3] = 1.0
            self.dye = [stored(resample(self.dye[0], dw, dh), self.storage), d2]
        if self.vel is None:
            self.vel = [np.zeros((sh, sw, 2), np.float32), np.zeros((sh, sw, 2), np.float32)]
        elif self.vel[0].shape[:2] != (sh, sw):
            self.vel = [stored(resample(self.vel[0], sw, sh), self.storage), np.zeros((sh, sw, 2), np.float32)]
        self.div = np.zeros((sh, sw), np.float32)
        self.curl = np.zeros((sh, sw), np.float32)
        self.prs = [np.zeros((sh, sw), np.float32), np.zeros((sh, sw), np.float32)]
        self.sim = (sw, sh)
        self.dyeres = (dw, dh)

    # script.js:1441-1462
    def splat(self, x, y, dx, dy, color):
        aspect = self.canvas[0] / self.canvas[1]
        radius = self.config["SPLAT_RADIUS"] / 100.0
        if aspect > 1:
            radius *= aspect
        a, r = f32(aspect), f32(radius)
        self.vel[0] = stored(splat(self.vel[0], f32(x), f32(y), a, r, (f32(dx), f32(dy), 0.0)), self.storage)
        self.dye[0] = stored(splat(self.dye[0], f32(x), f32(y), a, r, tuple(f32(c) for c in color)), self.storage)

    # script.js:1427-1439 + 1565-1571: five Math.random draws per splat: hue, x, y, dx, dy
    def multiple_splats(self, amount: int) -> List[List[float]]:
        log = []
        for _ in range(amount):
            c = [ch * 0.15 * 10.0 for ch in hsv_to_rgb(self.random(), 1.0, 1.0)]
            x = self.random()
            y = self.random()
            dx = 1000 * (self.random() - 0.5)
            dy = 1000 * (self.random() - 0.5)
            self.splat(x, y, dx, dy, c)
            log.append([x, y, dx, dy] + c)
        return log

    def params(self) -> Params:
        c = self.config
        return Params(f32(c["CURL"]), f32(c["PRESSURE"]), int(c["PRESSURE_ITERATIONS"]),
                      f32(c["VELOCITY_DISSIPATION"]), f32(c["DENSITY_DISSIPATION"]))

    # script.js:1231-1294 (one C call; fo_step swaps the pairs in place like DoubleFBO.swap)
    def step(self, dt: float = 0.016666, n: int = 1):
        sw, sh = self.sim
        dw, dh = self.dyeres
        P = self.params()
        for _ in range(n):
            vel = (FP * 2)(_p(self.vel[0]), _p(self.vel[1]))
            prs = (FP * 2)(_p(self.prs[0]), _p(self.prs[1]))
            dye = (FP * 2)(_p(self.dye[0]), _p(self.dye[1]))
            step_fn = lib().fo_step_f16 if self.storage == "f16" else lib().fo_step
            step_fn(sw, sh, dw, dh, vel, prs, dye, _p(self.div), _p(self.curl), f32(dt), C.byref(P))
            # read back which buffer is now "read" (pointer identity)
            for pair, arr in ((vel, self.vel), (prs, self.prs), (dye, self.dye)):
                if C.addressof(pair[0].contents) != arr[0].ctypes.data:
                    arr.reverse()

    def fields(self) -> Dict[str, np.ndarray]:
        return {"velocity": self.vel[0], "pressure": self.prs[0], "divergence": self.div,
                "curl": self.curl, "dye": self.dye[0]}


def num_threads() -> int:
    return int(lib().fo_num_threads())
