"""Host-side mirror of the reference's simulation surface over libfluid_hip.so.

The reference keeps its simulation API as page-level globals (script.js): `config` (59-85),
`initFramebuffers()` (982-1010), `splat()` (1441-1455), `multipleSplats()` (1427-1439),
`step(dt)` (1231-1294), `update()` (1176-1186), `framebufferToTexture()` (301-307) and the field
objects `velocity, dye, pressure, divergence, curl` (950-954).  `FluidSim` exposes the same names
with the same argument meaning, one instance per "page".  The JavaScript twin of this class is
webgl-fluid-simulation_amd/addon/fluid.js (Node N-API); both sit on the same C ABI.

All arithmetic happens in the HIP library; this file only holds the reference's host logic
(resolution rule, splat parameter rules, the Math.random call order, the dt clamp).
"""
from __future__ import annotations

import ctypes as C
import math
import random as _random
from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np

from . import _abi
from ._abi import FIELD_IDS, FluidError  # noqa: F401  (FluidError is re-exported)

# config keys the simulation path reads (script.js:59-85); display-only keys are not modelled
DEFAULT_CONFIG = {
    "SIM_RESOLUTION": 128,
    "DYE_RESOLUTION": 1024,
    "DENSITY_DISSIPATION": 1,
    "VELOCITY_DISSIPATION": 0.2,
    "PRESSURE": 0.8,
    "PRESSURE_ITERATIONS": 20,
    "CURL": 30,
    "SPLAT_RADIUS": 0.25,
    "SPLAT_FORCE": 6000,
    "COLORFUL": True,
    "COLOR_UPDATE_SPEED": 10,
    "PAUSED": False,
    # display compositor (render / captureScreenshot, script.js:61, 71-84)
    "CAPTURE_RESOLUTION": 512,
    "SHADING": True,
    "BACK_COLOR": {"r": 0, "g": 0, "b": 0},
    "TRANSPARENT": False,
    "BLOOM": True,
    "BLOOM_ITERATIONS": 8,
    "BLOOM_RESOLUTION": 256,
    "BLOOM_INTENSITY": 0.8,
    "BLOOM_THRESHOLD": 0.6,
    "BLOOM_SOFT_KNEE": 0.7,
    "SUNRAYS": True,
    "SUNRAYS_RESOLUTION": 196,
    "SUNRAYS_WEIGHT": 1.0,
}

SCHEDULES = {"passes": _abi.SCHED_PASSES, "fused": _abi.SCHED_FUSED}


def mulberry32(seed: int) -> Callable[[], float]:
    """Seedable stand-in for Math.random (the stream BASELINE.md's measurement plan names)."""
    state = seed & 0xFFFFFFFF

    def imul(a: int, b: int) -> int:
        return ((a & 0xFFFFFFFF) * (b & 0xFFFFFFFF)) & 0xFFFFFFFF

    def rnd() -> float:
        nonlocal state
        state = (state + 0x6D2B79F5) & 0xFFFFFFFF
        t = imul(state ^ (state >> 15), 1 | state)
        t = ((t + imul(t ^ (t >> 7), 61 | t)) & 0xFFFFFFFF) ^ t
        return ((t ^ (t >> 14)) & 0xFFFFFFFF) / 4294967296.0

    return rnd


def HSVtoRGB(h: float, s: float, v: float) -> Dict[str, float]:
    """script.js:1573-1595"""
    i = math.floor(h * 6)
    f = h * 6 - i
    p = v * (1 - s)
    q = v * (1 - f * s)
    t = v * (1 - (1 - f) * s)
    r, g, b = ((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))[int(i % 6)]
    return {"r": r, "g": g, "b": b}


def getResolution(resolution: float, drawing_w: int, drawing_h: int) -> Dict[str, int]:
    """script.js:1612-1624 (gl.drawingBufferWidth/Height = the canvas size)"""
    aspect = drawing_w / drawing_h
    if aspect < 1:
        aspect = 1.0 / aspect
    mn = int(math.floor(resolution + 0.5))           # Math.round
    mx = int(math.floor(resolution * aspect + 0.5))
    if drawing_w > drawing_h:
        return {"width": mx, "height": mn}
    return {"width": mn, "height": mx}


def _f32(x: float) -> float:
    """what gl.uniform1f does to a JS number"""
    return float(np.float32(x))


class Canvas:
    """stand-in for the page's <canvas> (only width/height are used by the simulation path)"""

    def __init__(self, width: int, height: int):
        self.width = int(width)
        self.height = int(height)


class FieldView:
    """width/height/texelSize view of one field, like the FBO objects of script.js:1064-1076"""

    def __init__(self, sim: "FluidSim", name: str, double: bool):
        self._sim, self._name, self._double = sim, name, double

    @property
    def width(self) -> int:
        return self._sim._info(self._name).width

    @property
    def height(self) -> int:
        return self._sim._info(self._name).height

    @property
    def texelSizeX(self) -> float:
        return 1.0 / self.width

    @property
    def texelSizeY(self) -> float:
        return 1.0 / self.height

    @property
    def read(self) -> "FieldView":
        # the library swaps read/write itself; the view always names the current read side
        if not self._double:
            raise AttributeError("%s is a single FBO" % self._name)
        return self


class Pointer:
    """pointerPrototype, script.js:87-98"""

    def __init__(self):
        self.id = -1
        self.texcoordX = self.texcoordY = self.prevTexcoordX = self.prevTexcoordY = 0.0
        self.deltaX = self.deltaY = 0.0
        self.down = self.moved = False
        self.color = {"r": 30, "g": 0, "b": 300}


class FluidSim:
    def __init__(self, canvas: Union[Canvas, Tuple[int, int]] = (512, 512), config: Optional[dict] = None,
                 device: int = 0, schedule: str = "fused", random: Optional[Callable[[], float]] = None, storage: str = "f32"):
        """storage: "f32" (what the headless reference the goldens come from keeps) or "f16" (what the reference's half-float
        textures hold on a real GPU, script.js:138: every pass output rounded to fp16, half the bytes per step)"""
        self._lib = _abi.lib()
        self.canvas = canvas if isinstance(canvas, Canvas) else Canvas(*canvas)
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.random = random or _random.random  # Math.random
        self.splatStack = []
        self.pointers = [Pointer()]          # script.js:100-102
        self.pixelRatio = 1.0
        self._colorUpdateTimer = 0.0
        self._device = device
        self._schedule = SCHEDULES[schedule]
        self._storage = _abi.STORAGE[storage]
        self._ctx = None
        self.initFramebuffers()
        self.velocity = FieldView(self, "velocity", True)
        self.dye = FieldView(self, "dye", True)
        self.pressure = FieldView(self, "pressure", True)
        self.divergence = FieldView(self, "divergence", False)
        self.curl = FieldView(self, "curl", False)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self._ctx is not None:
            self._lib.fluid_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, status: int):
        _abi.check(self._ctx, status)

    def _info(self, name: str) -> _abi.FieldInfo:
        fi = _abi.FieldInfo()
        self._check(self._lib.fluid_field_info_get(self._ctx, FIELD_IDS[name], C.byref(fi)))
        return fi

    # -- initFramebuffers(), script.js:982-1010 ---------------------------------------------
    def initFramebuffers(self):
        sim = getResolution(self.config["SIM_RESOLUTION"], self.canvas.width, self.canvas.height)
        dye = getResolution(self.config["DYE_RESOLUTION"], self.canvas.width, self.canvas.height)
        if self._ctx is None:
            d = _abi.Desc(sim["width"], sim["height"], dye["width"], dye["height"], self._device, 0, 1, 0, self._schedule, 0, 1, self._storage)
            ctx = C.c_void_p()
            rc = self._lib.fluid_create(C.byref(d), C.byref(ctx))
            if rc != _abi.FLUID_OK:
                _abi.check(None, rc)
            self._ctx = ctx
        else:
            self._check(self._lib.fluid_resize(self._ctx, sim["width"], sim["height"], dye["width"], dye["height"]))

    def set_schedule(self, schedule: str):
        self._schedule = SCHEDULES[schedule]
        self._check(self._lib.fluid_set_schedule(self._ctx, self._schedule))

    # -- splat(), script.js:1441-1462 ---------------------------------------------------------
    def correctRadius(self, radius: float) -> float:
        aspect = self.canvas.width / self.canvas.height
        if aspect > 1:
            radius *= aspect
        return radius

    def splat(self, x: float, y: float, dx: float, dy: float, color):
        if isinstance(color, dict):
            r, g, b = color["r"], color["g"], color["b"]
        else:
            r, g, b = color
        aspect = self.canvas.width / self.canvas.height
        radius = self.correctRadius(self.config["SPLAT_RADIUS"] / 100.0)
        self._check(self._lib.fluid_splat(self._ctx, x, y, dx, dy, r, g, b, aspect, radius))

    # -- generateColor / multipleSplats, script.js:1565-1571, 1427-1439 ------------------------
    def generateColor(self) -> Dict[str, float]:
        c = HSVtoRGB(self.random(), 1.0, 1.0)
        c["r"] *= 0.15
        c["g"] *= 0.15
        c["b"] *= 0.15
        return c

    def multipleSplats(self, amount: int):
        issued = []
        for _ in range(int(amount)):
            color = self.generateColor()
            color["r"] *= 10.0
            color["g"] *= 10.0
            color["b"] *= 10.0
            x = self.random()
            y = self.random()
            dx = 1000 * (self.random() - 0.5)
            dy = 1000 * (self.random() - 0.5)
            self.splat(x, y, dx, dy, color)
            issued.append([x, y, dx, dy, color["r"], color["g"], color["b"]])
        return issued

    # -- step(dt), script.js:1231-1294 -----------------------------------------------------------
    def params(self) -> _abi.Params:
        c = self.config
        return _abi.Params(c["CURL"], c["PRESSURE"], int(c["PRESSURE_ITERATIONS"]),
                           c["VELOCITY_DISSIPATION"], c["DENSITY_DISSIPATION"])

    def step(self, dt: float, n: int = 1):
        P = self.params()
        self._check(self._lib.fluid_step_n(self._ctx, int(n), dt, C.byref(P)))

    # -- update(), script.js:1176-1186, without the render: dt clamp (1191), colours, inputs, PAUSED gate ----
    def update(self, wall_dt: float):
        dt = min(wall_dt, 0.016666)
        self.updateColors(dt)
        self.applyInputs()
        if not self.config["PAUSED"]:
            self.step(dt)
        return dt

    # -- input path: the reference's listeners, headless ---------------------------------------
    def updateColors(self, dt: float):                       # script.js:1207-1217
        if not self.config["COLORFUL"]:
            return
        self._colorUpdateTimer += dt * self.config["COLOR_UPDATE_SPEED"]
        if self._colorUpdateTimer >= 1:
            self._colorUpdateTimer = math.fmod(self._colorUpdateTimer, 1.0)   # wrap(value, 0, 1), script.js:1603-1607
            for p in self.pointers:
                p.color = self.generateColor()

    def applyInputs(self):                                   # script.js:1219-1229
        if self.splatStack:
            self.multipleSplats(self.splatStack.pop())
        for p in self.pointers:
            if p.moved:
                p.moved = False
                self.splatPointer(p)

    def splatPointer(self, pointer):                         # script.js:1421-1425
        dx = pointer.deltaX * self.config["SPLAT_FORCE"]
        dy = pointer.deltaY * self.config["SPLAT_FORCE"]
        self.splat(pointer.texcoordX, pointer.texcoordY, dx, dy, pointer.color)

    def scaleByPixelRatio(self, value: float) -> int:        # script.js:1626-1629
        return math.floor(value * self.pixelRatio)

    def correctDeltaX(self, delta: float) -> float:          # script.js:1560-1564
        aspect = self.canvas.width / self.canvas.height
        return delta * aspect if aspect < 1 else delta

    def correctDeltaY(self, delta: float) -> float:          # script.js:1566-1570
        aspect = self.canvas.width / self.canvas.height
        return delta / aspect if aspect > 1 else delta

    def updatePointerDownData(self, pointer, pid, posX, posY):   # script.js:1532-1543
        pointer.id = pid
        pointer.down = True
        pointer.moved = False
        pointer.texcoordX = posX / self.canvas.width
        pointer.texcoordY = 1.0 - posY / self.canvas.height
        pointer.prevTexcoordX = pointer.texcoordX
        pointer.prevTexcoordY = pointer.texcoordY
        pointer.deltaX = 0
        pointer.deltaY = 0
        pointer.color = self.generateColor()

    def updatePointerMoveData(self, pointer, posX, posY):        # script.js:1545-1553
        pointer.prevTexcoordX = pointer.texcoordX
        pointer.prevTexcoordY = pointer.texcoordY
        pointer.texcoordX = posX / self.canvas.width
        pointer.texcoordY = 1.0 - posY / self.canvas.height
        pointer.deltaX = self.correctDeltaX(pointer.texcoordX - pointer.prevTexcoordX)
        pointer.deltaY = self.correctDeltaY(pointer.texcoordY - pointer.prevTexcoordY)
        pointer.moved = abs(pointer.deltaX) > 0 or abs(pointer.deltaY) > 0

    def dispatch(self, e: dict):
        """one recorded DOM event, through the bodies of the reference's listeners (script.js:1464-1530)"""
        t, ptrs = e["type"], self.pointers
        if t == "mousedown":
            pointer = next((p for p in ptrs if p.id == -1), None) or Pointer()
            self.updatePointerDownData(pointer, -1, self.scaleByPixelRatio(e["offsetX"]), self.scaleByPixelRatio(e["offsetY"]))
        elif t == "mousemove":
            if ptrs[0].down:
                self.updatePointerMoveData(ptrs[0], self.scaleByPixelRatio(e["offsetX"]), self.scaleByPixelRatio(e["offsetY"]))
        elif t == "mouseup":
            ptrs[0].down = False
        elif t == "touchstart":
            touches = e["touches"]
            while len(touches) >= len(ptrs):
                ptrs.append(Pointer())
            for i, tc in enumerate(touches):
                self.updatePointerDownData(ptrs[i + 1], tc["identifier"], self.scaleByPixelRatio(tc["pageX"]), self.scaleByPixelRatio(tc["pageY"]))
        elif t == "touchmove":
            for i, tc in enumerate(e["touches"]):
                if ptrs[i + 1].down:
                    self.updatePointerMoveData(ptrs[i + 1], self.scaleByPixelRatio(tc["pageX"]), self.scaleByPixelRatio(tc["pageY"]))
        elif t == "touchend":
            for tc in e["touches"]:
                pointer = next((p for p in ptrs if p.id == tc["identifier"]), None)
                if pointer is not None:
                    pointer.down = False
        elif t == "keydown":
            if e.get("code") == "KeyP":
                self.config["PAUSED"] = not self.config["PAUSED"]
            if e.get("key") == " ":
                self.splatStack.append(int(self.random() * 20) + 5)
        else:
            raise ValueError("unknown event type %r" % t)

    def replay(self, frames):
        """frames = [{dt, events}]: dispatch the frame's events, then one update() with its dt"""
        for f in frames:
            for e in f.get("events", []):
                self.dispatch(e)
            self.update(f["dt"])

    def sync(self):
        self._check(self._lib.fluid_sync(self._ctx))

    # -- display compositor: render(target) / captureScreenshot(), script.js:287-349, 1296-1419 ------
    def setDitheringTexture(self, r: np.ndarray):
        """the R channel (0..1) of the dithering texture (script.js:958); default: the reference's 1 x 1 white placeholder"""
        a = np.ascontiguousarray(r, np.float32)
        self._check(self._lib.fluid_set_dither(self._ctx, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0]))

    def _display_params(self) -> "_abi.DisplayParams":
        c = self.config
        bloom = getResolution(c["BLOOM_RESOLUTION"], self.canvas.width, self.canvas.height)
        sun = getResolution(c["SUNRAYS_RESOLUTION"], self.canvas.width, self.canvas.height)
        bc = c["BACK_COLOR"]                                     # normalizeColor, script.js:1597-1604
        return _abi.DisplayParams(int(bool(c["SHADING"])), int(bool(c["BLOOM"])), int(bool(c["SUNRAYS"])), int(bool(c["TRANSPARENT"])),
                                  bc["r"] / 255, bc["g"] / 255, bc["b"] / 255, bloom["width"], bloom["height"], int(c["BLOOM_ITERATIONS"]),
                                  float(c["BLOOM_INTENSITY"]), float(c["BLOOM_THRESHOLD"]), float(c["BLOOM_SOFT_KNEE"]),
                                  sun["width"], sun["height"], float(c["SUNRAYS_WEIGHT"]))

    def render(self, width: int, height: int) -> np.ndarray:
        """render(target) into a width x height float target; returns framebufferToTexture(target): [h, w, 4], row 0 = bottom"""
        p = self._display_params()
        self._check(self._lib.fluid_render(self._ctx, int(width), int(height), C.byref(p)))
        out = np.empty((height, width, 4), np.float32)
        self._check(self._lib.fluid_read_frame(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def captureScreenshot(self) -> np.ndarray:
        """captureScreenshot() up to the PNG encoder: the RGBA8 image [h, w, 4], top row first (normalizeTexture)"""
        res = getResolution(self.config["CAPTURE_RESOLUTION"], self.canvas.width, self.canvas.height)
        self.render(res["width"], res["height"])
        out = np.empty((res["height"], res["width"], 4), np.uint8)
        self._check(self._lib.fluid_read_frame_rgba8(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def display_buffer(self, which: str) -> np.ndarray:
        """the bloom ([h, w, 4]) or blurred sunrays ([h, w]) buffer of the last render"""
        wid = {"bloom": _abi.DISPLAY_BLOOM, "sunrays": _abi.DISPLAY_SUNRAYS}[which]
        w, h = C.c_int(0), C.c_int(0)
        self._check(self._lib.fluid_read_display_buffer(self._ctx, wid, None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, 4) if which == "bloom" else (h.value, w.value), np.float32)
        self._check(self._lib.fluid_read_display_buffer(self._ctx, wid, out.ctypes.data_as(C.c_void_p), out.nbytes, None, None))
        return out

    # -- field access ------------------------------------------------------------------------------
    def read(self, name: str) -> np.ndarray:
        """field in its native channel count: velocity [H,W,2], dye [H,W,4], others [H,W]; row 0 = bottom"""
        fi = self._info(name)
        shape = (fi.rows, fi.width) if fi.channels == 1 else (fi.rows, fi.width, fi.channels)
        out = np.empty(shape, np.float32)
        self._check(self._lib.fluid_read_field(self._ctx, FIELD_IDS[name], out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def write(self, name: str, arr: np.ndarray):
        fi = self._info(name)
        shape = (fi.rows, fi.width) if fi.channels == 1 else (fi.rows, fi.width, fi.channels)
        a = np.ascontiguousarray(arr, dtype=np.float32)
        if a.shape != shape:
            raise ValueError("%s expects shape %s, got %s" % (name, shape, a.shape))
        self._check(self._lib.fluid_write_field(self._ctx, FIELD_IDS[name], a.ctypes.data_as(C.c_void_p), a.nbytes))

    def framebufferToTexture(self, target: Union[str, FieldView]) -> np.ndarray:
        """script.js:301-307: readPixels(RGBA, FLOAT) — R and RG fields come back padded to (r, g, 0, 1)"""
        name = target._name if isinstance(target, FieldView) else target
        a = self.read(name)
        if a.ndim == 2:
            a = a[..., None]
        h, w, nc = a.shape
        out = np.zeros((h, w, 4), np.float32)
        out[..., 3] = 1.0
        out[..., :nc] = a
        return out.reshape(-1)

    def fields(self) -> Dict[str, np.ndarray]:
        return {k: self.read(k) for k in FIELD_IDS}

    def device_view(self, name: str, stream=None):
        """torch tensor [rows, width, channels] aliasing the field's CURRENT read buffer on the device (zero copy through
        __cuda_array_interface__; the padding columns of the pitch are sliced off).

        ORDERED for the caller (include/fluid_hip.h, fluid_field_device_ptr's rule (2)): the torch stream that is current when this is
        called (or `stream`) waits ON THE DEVICE for everything the solver has enqueued so far — steps still running, the packed dye's
        conversion back to RGBA — so torch work issued on that stream afterwards reads the finished field.  No host synchronisation.
        (Round 4 left this to the caller's sim.sync() and then enqueued the conversion BEHIND that sync: BENCH_r04's mismatch.)

        READ-ONLY and short-lived: the pointer is the buffer that is current NOW — any step, splat, pass or write swaps the field's
        ping-pong buffers (and may free them: resize), after which the view shows the spare buffer or dangles.  Torch work that still
        reads the view when the solver is next called must be ordered in front of it: `sim.wait_for_torch()`."""
        import torch
        ptr = C.c_void_p()
        self._check(self._lib.fluid_field_device_ptr(self._ctx, FIELD_IDS[name], C.byref(ptr)))
        fi = self._info(name)
        # a FluidSim is a whole-domain context: no ghost rows / columns, array column 0 = global column 0 (a stripe or tile context has its
        # own view with the ghost geometry: fluid_hip.stripes.HipStripeEngine.view)
        if fi.halo or fi.halo_x or fi.array_col0 or fi.cols != fi.width or fi.rows != fi.height:
            raise FluidError(_abi.ERR_UNSUPPORTED, "device_view is for whole-domain contexts")
        with torch.cuda.device(self._device):
            s = torch.cuda.current_stream() if stream is None else stream
            self._check(self._lib.fluid_stream_wait_context(self._ctx, C.c_void_p(s.cuda_stream)))
        shape = (fi.rows + 2 * fi.halo, fi.pitch, fi.channels)

        class _DeviceArray:  # minimal CUDA-array-interface carrier
            __cuda_array_interface__ = {"shape": shape, "typestr": "<f%d" % fi.bytes_per_channel, "data": (ptr.value, False), "version": 2}

        return torch.as_tensor(_DeviceArray(), device="cuda:%d" % self._device)[:, :fi.width]

    def wait_for_torch(self, stream=None):
        """the solver's next call runs behind everything enqueued on torch's current stream (or `stream`) so far: call it between the last
        torch op that reads a device_view and the next step / splat / write (fluid_context_wait_stream; no host synchronisation)"""
        import torch
        with torch.cuda.device(self._device):
            s = torch.cuda.current_stream() if stream is None else stream
            self._check(self._lib.fluid_context_wait_stream(self._ctx, C.c_void_p(s.cuda_stream)))

    # -- single passes (test / stripe-driver port) -------------------------------------------------
    def run_pass(self, name: str, dt: float = 0.016666, iters: int = 1, ext: int = 0):
        L, c, P = self._lib, self._ctx, self.params()
        if name == "curl":
            rc = L.fluid_pass_curl(c, ext)
        elif name == "vorticity":
            rc = L.fluid_pass_vorticity(c, P.curl, dt, ext)
        elif name == "divergence":
            rc = L.fluid_pass_divergence(c, ext)
        elif name == "curl_vorticity_divergence":
            rc = L.fluid_pass_curl_vorticity_divergence(c, P.curl, dt, ext)
        elif name == "clear":
            rc = L.fluid_pass_clear(c, P.pressure, ext)
        elif name == "jacobi":
            rc = L.fluid_pass_jacobi(c, iters, ext)
        elif name == "clear_jacobi":
            rc = L.fluid_pass_clear_jacobi(c, P.pressure, iters, ext)
        elif name == "gradsub":
            rc = L.fluid_pass_gradsub(c, ext)
        elif name == "advect_velocity":
            rc = L.fluid_pass_advect_velocity(c, dt, P.velocity_dissipation, ext)
        elif name == "advect_dye":
            rc = L.fluid_pass_advect_dye(c, dt, P.density_dissipation)
        elif name == "advect":
            rc = L.fluid_pass_advect(c, dt, P.velocity_dissipation, P.density_dissipation)
        else:
            raise ValueError("unknown pass " + name)
        self._check(rc)

    def set_curl_output(self, on: bool):
        """fluid_set_curl_output: off = no step stores its curl field (the reference reads it inside step() only, script.js:1239-1243);
        read("curl") then raises until the output is on again and a step has run"""
        self._check(self._lib.fluid_set_curl_output(self._ctx, 1 if on else 0))

    # -- timing ------------------------------------------------------------------------------------
    def set_timing(self, on: bool):
        self._check(self._lib.fluid_set_timing(self._ctx, 1 if on else 0))

    def timings(self) -> Dict[str, float]:
        t = _abi.Timings()
        self._check(self._lib.fluid_get_timings(self._ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _abi.Timings._fields_}

    def set_step_marks(self, capacity: int):
        """events between the first `capacity` steps of every following step(dt, n) call; nothing waits for them (0 = off)"""
        self._check(self._lib.fluid_set_step_marks(self._ctx, int(capacity)))

    def step_marks(self) -> list:
        """device milliseconds of each marked step of the last step(dt, n) call (waits for that call's last mark)"""
        n = C.c_int(0)
        self._check(self._lib.fluid_get_step_marks(self._ctx, None, 0, C.byref(n)))
        buf = (C.c_float * max(n.value, 1))()
        self._check(self._lib.fluid_get_step_marks(self._ctx, buf, n.value, C.byref(n)))
        return [float(buf[k]) for k in range(n.value)]

    def schedule_info(self, n_steps: int = 1, dt: float = 0.016666) -> Dict[str, int]:
        """which kernels step(dt, n_steps) would launch on this context right now (fluid_schedule_info_get)"""
        P = self.params()
        info = _abi.ScheduleInfo()
        self._check(self._lib.fluid_schedule_info_get(self._ctx, int(n_steps), dt, C.byref(P), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _abi.ScheduleInfo._fields_}
