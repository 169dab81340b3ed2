"""ctypes binding of the C ABI in include/fluid_hip.h (libfluid_hip.so).

No fallback: if the shared library is missing or no HIP device is visible, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# FLUID_HIP_LIB: load another build of the same library (kernel A/B experiments); the default is the in-tree build
LIB_PATH = os.environ.get("FLUID_HIP_LIB") or os.path.join(PKG_DIR, "libfluid_hip.so")
# the lab build (make PROBES=1): every tile shape / variant of profiles/ and the FLUID_* knobs that select them; the product reads no knob
PROBES_LIB_PATH = os.path.join(PKG_DIR, "libfluid_hip_probes.so")

FLUID_OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_OOM, ERR_HALO, ERR_UNSUPPORTED, ERR_COMM = -1, -2, -3, -4, -5, -6, -7
VELOCITY, PRESSURE, DIVERGENCE, CURL, DYE = 0, 1, 2, 3, 4
FIELD_IDS = {"velocity": VELOCITY, "pressure": PRESSURE, "divergence": DIVERGENCE, "curl": CURL, "dye": DYE}
FIELD_CHANNELS = {VELOCITY: 2, PRESSURE: 1, DIVERGENCE: 1, CURL: 1, DYE: 4}
SCHED_PASSES, SCHED_FUSED = 0, 1


class FluidError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("libfluid_hip: %s (status %d)" % (message, status))
        self.status = status


STORE_F32, STORE_F16 = 0, 1
STORAGE = {"f32": STORE_F32, "f16": STORE_F16}   # fluid_storage


class Desc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("sim_w", "sim_h", "dye_w", "dye_h", "device", "part", "parts", "halo", "schedule", "part_x", "parts_x", "storage")]


class Params(C.Structure):
    _fields_ = [("curl", C.c_float), ("pressure", C.c_float), ("iterations", C.c_int),
                ("velocity_dissipation", C.c_float), ("density_dissipation", C.c_float)]


class FieldInfo(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("width", "height", "channels", "row0", "rows", "halo", "col0", "cols", "halo_x", "bytes_per_channel", "pitch", "array_col0")]


class Timings(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("curl_ms", "vorticity_ms", "divergence_ms", "clear_ms", "jacobi_ms", "gradsub_ms",
                                         "advect_velocity_ms", "advect_dye_ms", "total_ms")] + \
               [("jacobi_launches", C.c_int), ("steps", C.c_int), ("folded_launches", C.c_int)]


class DisplayParams(C.Structure):
    _fields_ = [("shading", C.c_int), ("bloom", C.c_int), ("sunrays", C.c_int), ("transparent", C.c_int),
                ("back_r", C.c_float), ("back_g", C.c_float), ("back_b", C.c_float),
                ("bloom_w", C.c_int), ("bloom_h", C.c_int), ("bloom_iterations", C.c_int),
                ("bloom_intensity", C.c_double), ("bloom_threshold", C.c_double), ("bloom_soft_knee", C.c_double),
                ("sunrays_w", C.c_int), ("sunrays_h", C.c_int), ("sunrays_weight", C.c_double)]


DISPLAY_BLOOM, DISPLAY_SUNRAYS = 0, 1


class ScheduleInfo(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("fused", "jacobi_shape", "jacobi_launches", "gradsub_folded", "chained", "curl_stores", "launches", "runs_ahead", "pending_adopted", "dye_packed", "jacobi_chained")]


class StripeOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("iters", C.c_int), ("ext", C.c_int), ("n_items", C.c_int), ("field", C.c_int * 2), ("rows", C.c_int * 2)]


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


OP_EXCHANGE, OP_CURL_VORT_DIV, OP_CLEAR, OP_CLEAR_JACOBI, OP_JACOBI, OP_GRADSUB, OP_ADVECT = range(7)

# every symbol include/fluid_hip.h declares: name -> (restype, argtypes)
_CTX = C.c_void_p
_F = C.c_float
_I = C.c_int
SYMBOLS = {
    "fluid_abi_version": (_I, []),
    "fluid_build_flavor": (C.c_char_p, []),
    "fluid_error_string": (C.c_char_p, [_I]),
    "fluid_last_error": (C.c_char_p, [_CTX]),
    "fluid_device_count": (_I, [C.POINTER(_I)]),
    "fluid_create": (_I, [C.POINTER(Desc), C.POINTER(_CTX)]),
    "fluid_destroy": (_I, [_CTX]),
    "fluid_resize": (_I, [_CTX, _I, _I, _I, _I]),
    "fluid_set_schedule": (_I, [_CTX, _I]),
    "fluid_set_stream": (_I, [_CTX, C.c_void_p, _I]),
    "fluid_splat": (_I, [_CTX] + [_F] * 9),
    "fluid_step": (_I, [_CTX, _F, C.POINTER(Params)]),
    "fluid_step_n": (_I, [_CTX, _I, _F, C.POINTER(Params)]),
    "fluid_sync": (_I, [_CTX]),
    "fluid_read_field": (_I, [_CTX, _I, C.c_void_p, C.c_size_t]),
    "fluid_write_field": (_I, [_CTX, _I, C.c_void_p, C.c_size_t]),
    "fluid_field_info_get": (_I, [_CTX, _I, C.POINTER(FieldInfo)]),
    "fluid_pass_curl": (_I, [_CTX, _I]),
    "fluid_pass_vorticity": (_I, [_CTX, _F, _F, _I]),
    "fluid_pass_divergence": (_I, [_CTX, _I]),
    "fluid_pass_curl_vorticity_divergence": (_I, [_CTX, _F, _F, _I]),
    "fluid_pass_clear": (_I, [_CTX, _F, _I]),
    "fluid_pass_jacobi": (_I, [_CTX, _I, _I]),
    "fluid_pass_clear_jacobi": (_I, [_CTX, _F, _I, _I]),
    "fluid_pass_gradsub": (_I, [_CTX, _I]),
    "fluid_pass_advect_velocity": (_I, [_CTX, _F, _F, _I]),
    "fluid_pass_advect_dye": (_I, [_CTX, _F, _F]),
    "fluid_pass_advect": (_I, [_CTX, _F, _F, _F]),
    "fluid_pass_splat": (_I, [_CTX, _I] + [_F] * 7),
    "fluid_halo_pack": (_I, [_CTX, _I, _I, _I, C.c_void_p]),
    "fluid_halo_unpack": (_I, [_CTX, _I, _I, _I, C.c_void_p]),
    "fluid_field_device_ptr": (_I, [_CTX, _I, C.POINTER(C.c_void_p)]),
    "fluid_stream_wait_context": (_I, [_CTX, C.c_void_p]),
    "fluid_context_wait_stream": (_I, [_CTX, C.c_void_p]),
    "fluid_halo_check": (_I, [_CTX]),
    "fluid_stripe_plan": (_I, [_I, _I, _I, _I, _I, C.POINTER(StripeOp), _I, C.POINTER(_I)]),
    "fluid_set_reach": (_I, [_CTX, _I]),
    "fluid_advect_exchange_rows": (_I, [_CTX, C.POINTER(_I), C.POINTER(_I)]),
    "fluid_set_overlap": (_I, [_CTX, _I]),
    "fluid_set_link_model": (_I, [_CTX, _F, _F]),
    "fluid_comm_calibrate_link": (_I, [_CTX, _I, C.POINTER(_F), C.POINTER(_F)]),
    "fluid_comm_set_library": (_I, [C.c_char_p]),
    "fluid_comm_unique_id": (_I, [C.POINTER(CommId)]),
    "fluid_comm_init": (_I, [_CTX, C.POINTER(CommId)]),
    "fluid_comm_selftest": (_I, [_CTX, _I]),
    "fluid_exchange_count": (C.c_long, [_CTX]),
    "fluid_group_step_n": (_I, [C.POINTER(_CTX), _I, _I, _F, C.POINTER(Params)]),
    "fluid_set_dither": (_I, [_CTX, C.c_void_p, _I, _I]),
    "fluid_render": (_I, [_CTX, _I, _I, C.POINTER(DisplayParams)]),
    "fluid_read_frame": (_I, [_CTX, C.c_void_p, C.c_size_t]),
    "fluid_read_frame_rgba8": (_I, [_CTX, C.c_void_p, C.c_size_t]),
    "fluid_read_display_buffer": (_I, [_CTX, _I, C.c_void_p, C.c_size_t, C.POINTER(_I), C.POINTER(_I)]),
    "fluid_set_curl_output": (_I, [_CTX, _I]),
    "fluid_set_timing": (_I, [_CTX, _I]),
    "fluid_get_timings": (_I, [_CTX, C.POINTER(Timings)]),
    "fluid_schedule_info_get": (_I, [_CTX, _I, _F, C.POINTER(Params), C.POINTER(ScheduleInfo)]),
    "fluid_set_step_marks": (_I, [_CTX, _I]),
    "fluid_get_step_marks": (_I, [_CTX, C.POINTER(_F), _I, C.POINTER(_I)]),
}

_lib = None


def build(force: bool = False) -> str:
    """(Re)build libfluid_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", PKG_DIR, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  If libfluid_hip.so is loaded
    first it binds /opt/rocm's copies, torch then maps its bundled ones as well, and the process ends up with
    TWO HIP runtimes that cannot share streams, events or device pointers (torch then even fails with "No HIP
    GPUs are available").  Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to the
    copy that is already mapped, so kernels, torch tensors and RCCL all live in one runtime.
    FLUID_HIP_NO_TORCH=1 skips this (pure ctypes / non-torch hosts)."""
    if os.environ.get("FLUID_HIP_NO_TORCH") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    """Load libfluid_hip.so and bind every declared symbol; raises if the library is absent."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise FluidError(ERR_UNSUPPORTED, "%s not built (run `make -C %s`); there is no CPU fallback" % (LIB_PATH, PKG_DIR))
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        if L.fluid_abi_version() != 10:
            raise FluidError(ERR_UNSUPPORTED, "ABI version mismatch")
        _lib = L
        _point_at_torch_rccl(L)
    return _lib


def _point_at_torch_rccl(L):
    """If torch is in the process, its bundled librccl.so is (or will be) mapped next to its bundled HIP runtime:
    hand that file to libfluid_hip so both use ONE RCCL on ONE runtime.  Otherwise the system librccl.so.1."""
    import sys
    t = sys.modules.get("torch")
    if t is None or os.environ.get("FLUID_RCCL_LIB"):
        return
    cand = os.path.join(os.path.dirname(t.__file__), "lib", "librccl.so")
    if os.path.exists(cand):
        L.fluid_comm_set_library(cand.encode())


def stripe_plan(halo: int, dye_halo: int, iterations: int, advect_rows: int = None, advect_dye_rows: int = None):
    """the native per-step plan as a list of tuples: ("exchange", [(field, rows), ...]) or (kind, iters, ext);
    advect_rows / advect_dye_rows default to the full ghost depth (what the hosted driver exchanges)"""
    L = lib()
    n = C.c_int(0)
    va = halo if advect_rows is None else advect_rows
    vd = dye_halo if advect_dye_rows is None else advect_dye_rows
    check(None, L.fluid_stripe_plan(halo, dye_halo, iterations, va, vd, None, 0, C.byref(n)))
    ops = (StripeOp * n.value)()
    check(None, L.fluid_stripe_plan(halo, dye_halo, iterations, va, vd, ops, n.value, C.byref(n)))
    names = {OP_CURL_VORT_DIV: "curl_vorticity_divergence", OP_CLEAR: "clear", OP_CLEAR_JACOBI: "clear_jacobi",
             OP_JACOBI: "jacobi", OP_GRADSUB: "gradsub", OP_ADVECT: "advect"}
    fields = {v: k for k, v in FIELD_IDS.items()}
    out = []
    for op in ops:
        if op.kind == OP_EXCHANGE:
            out.append(("exchange", [(fields[op.field[i]], op.rows[i]) for i in range(op.n_items)]))
        else:
            out.append((names[op.kind], op.iters, op.ext))
    return out


def check(ctx, status: int):
    if status != FLUID_OK:
        L = lib()
        detail = L.fluid_last_error(ctx) or b""
        raise FluidError(status, "%s: %s" % (L.fluid_error_string(status).decode(), detail.decode()))


def device_count() -> int:
    n = C.c_int(0)
    lib().fluid_device_count(C.byref(n))
    return n.value
