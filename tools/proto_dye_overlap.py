#!/usr/bin/env python3
"""Prototype (timing only, results are racy): the dye advection of step n on a second, CU-masked HIP stream under the
curl/vorticity/divergence + Jacobi + gradient-subtract kernels of step n + 1.  Usage: tools/proto_dye_overlap.py [dye_cus ...]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666


def masked_stream(hip, bits):
    words = (C.c_uint32 * 8)(*[0] * 8)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return s


def main():
    import torch  # noqa: F401  (one HIP runtime)
    import fluid_hip
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    for fn in ("hipEventCreateWithFlags", "hipEventRecord", "hipStreamWaitEvent", "hipStreamSynchronize"):
        getattr(hip, fn).restype = C.c_int
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    size, iters, steps = 4096, 50, 100
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    out = {}
    for dye_cus in [int(a) for a in sys.argv[1:]] or [0, 16, 32, 48, 64]:
        sim = fluid_hip.FluidSim(canvas=(size, size), config=cfg, random=fluid_hip.mulberry32(1234))
        sim.multipleSplats(20)
        L, ctx, P = sim._lib, sim._ctx, sim.params()
        if dye_cus == 0:   # reference: the fused step on the context's own stream
            sim.step(DT, 20); sim.sync()
            t0 = time.perf_counter(); sim.step(DT, steps); sim.sync()
            out["fused_step_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
            sim.close()
            continue
        main_s = masked_stream(hip, range(0, 256 - dye_cus))
        dye_s = masked_stream(hip, range(256 - dye_cus, 256))
        ev_vel, ev_dye = C.c_void_p(), C.c_void_p()
        hip.hipEventCreateWithFlags(C.byref(ev_vel), 2)
        hip.hipEventCreateWithFlags(C.byref(ev_dye), 2)

        def one_step(overlap=True):
            L.fluid_set_stream(ctx, main_s, 1)
            L.fluid_pass_curl_vorticity_divergence(ctx, P.curl, DT, 0)
            L.fluid_pass_clear_jacobi(ctx, P.pressure, iters, 0)
            L.fluid_pass_gradsub(ctx, 0)
            L.fluid_pass_advect_velocity(ctx, DT, P.velocity_dissipation, 0)
            if overlap:
                hip.hipEventRecord(ev_vel, main_s)
                hip.hipStreamWaitEvent(dye_s, ev_vel, 0)
                L.fluid_set_stream(ctx, dye_s, 1)
            L.fluid_pass_advect_dye(ctx, DT, P.density_dissipation)

        for mode in (True, False):
            for _ in range(20):
                one_step(mode)
            hip.hipStreamSynchronize(main_s); hip.hipStreamSynchronize(dye_s)
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step(mode)
            hip.hipStreamSynchronize(main_s); hip.hipStreamSynchronize(dye_s)
            out["cus%d_%s_ms" % (dye_cus, "overlap" if mode else "serial_split")] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        L.fluid_set_stream(ctx, None, 0)
        sim.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
