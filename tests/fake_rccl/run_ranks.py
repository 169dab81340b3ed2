"""TEST INFRASTRUCTURE — runs N stripe "ranks" as threads of this process, each stepping its stripe context through
fluid_step_n (the native plan + ncclSend / ncclRecv of csrc/fluid_stripes.cpp) with the in-process RCCL stand-in
loaded in place of librccl.so (FLUID_RCCL_LIB), and compares the assembled fields bit for bit with the single domain.
Usage: run_ranks.py '<json args>' (spawned by tests/test_stripes_gpu.py)"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))


def main():
    a = json.loads(sys.argv[1])
    import fluid_hip
    from fluid_hip import _abi
    from fluid_hip.sim import getResolution
    from fluid_hip.stripes import HipStripeEngine, new_comm_id
    world, halo, steps, cfg, canvas = a["world"], a["halo"], a["steps"], a["config"], tuple(a["canvas"])
    tx = a.get("tiles_x", 1)
    ty = world // tx
    storage = a.get("storage", "f32")
    full = dict(fluid_hip.DEFAULT_CONFIG, **cfg)
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, random=fluid_hip.mulberry32(9), storage=storage) as one:
        splats = one.multipleSplats(6)
        one.step(0.016666, steps)
        if a.get("lone_touch"):      # the ranks step a second time, behind something ONE of them does alone in between
            one.step(0.016666, steps)
        want = one.fields()
    sim = getResolution(full["SIM_RESOLUTION"], *canvas)
    dye = getResolution(full["DYE_RESOLUTION"], *canvas)
    cid = new_comm_id()
    out, errs, links, packed = [None] * world, [], [None] * world, []
    aspect = canvas[0] / canvas[1]
    radius = full["SPLAT_RADIUS"] / 100.0 * (aspect if aspect > 1 else 1.0)

    def rank(r):
        try:
            e = HipStripeEngine((sim["width"], sim["height"]), (dye["width"], dye["height"]), r // tx, ty, halo, _abi.SCHED_FUSED, 0,
                                part_x=r % tx, parts_x=tx, storage=storage)
            e.use_own_stream()
            if "overlap" in a:
                e.set_overlap(a["overlap"])
            e.comm_init(cid)                       # collective: blocks until every rank thread has called it
            if a.get("calibrate"):                 # the start-up link probe between real rank threads (collective as well): it must leave the
                links[r] = e.calibrate_link(5)     # communicator in step with its neighbours' and return a model
            for x, y, dx, dy, cr, cg, cb in splats:
                e.splat(x, y, dx, dy, cr, cg, cb, aspect, radius)
            e.step_n(steps, 0.016666, full)
            packed.append(bool(e.schedule_info(steps, 0.016666, full)["dye_packed"]))
            lt = a.get("lone_touch")
            if lt:
                # ONE rank alone does something that takes the dye's known alpha away from it (a raw device pointer: whatever is written
                # through it the context does not see) or makes it ineligible; its neighbours know nothing of it.  The set must agree on the
                # wire format of the next call by itself (fluid_stripes.cpp dye_format_agree) and still leave the single domain's bits
                if r == lt["rank"]:
                    import ctypes as C
                    ptr = C.c_void_p()
                    e._ck(e.lib.fluid_field_device_ptr(e.ctx, _abi.FIELD_IDS["dye"], C.byref(ptr)))
                e.step_n(steps, 0.016666, full)
                packed.append(bool(e.schedule_info(steps, 0.016666, full)["dye_packed"]))
            e.sync()
            e.check_halo()
            out[r] = ({k: e.read(k) for k in ("velocity", "pressure", "divergence", "curl", "dye")}, e.exchange_count())
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    if errs or any(o is None for o in out):
        print(json.dumps({"ok": False, "errors": errs}))
        return
    def assemble(k):
        return np.concatenate([np.concatenate([out[y * tx + x][0][k] for x in range(tx)], axis=1) for y in range(ty)], axis=0)
    bad = [k for k in want if not np.array_equal(assemble(k), want[k])]
    print(json.dumps({"ok": not bad, "mismatch": bad, "exchanges": out[0][1], "rccl": os.environ.get("FLUID_RCCL_LIB"), "links": links, "packed": packed}))


if __name__ == "__main__":
    main()
