#!/usr/bin/env python3
"""The zero-copy ordering contract of fluid_field_device_ptr (include/fluid_hip.h), exercised the two ways a consumer gets it wrong or right.

BENCH_r04.json was an error record: `fused_vs_passes_4096: MISMATCH`.  The bench did `sim.step(); sim.sync()` and THEN asked for the dye's
device pointer; for a fused 4096^2 context the dye is packed to three floats per texel, so that call enqueued the conversion back to RGBA
(k_dye_unpack, 470 MB) on the context's non-blocking stream and returned the pointer at once; torch.equal read it on the null stream.

  legacy   : step, sync, raw fluid_field_device_ptr, compare at once on torch's stream       (round 4's bench sequence)
  unsynced : queue >= 50 ms of steps, NO sync, FluidSim.device_view, compare at once         (what the ordered device_view promises)
  unsynced_velocity : the same on the velocity field — nothing to convert, so fluid_field_device_ptr does not wait on the host and the
             ONLY thing between the queued steps and torch's read is the event of fluid_stream_wait_context

Each trial compares the fused context's dye with a per-pass context that ran the same calls and was read through the synchronous host
path.  `--tree DIR` loads the package (and its libfluid_hip.so) from another checkout — build_ab/r04tree is `git archive d63ebbf`, the
tree the driver benched: there `legacy` must FAIL some of the time and `unsynced` every time, or the diagnosis is wrong.

    python tools/device_view_race.py --tree build_ab/r04tree --trials 50     # round 4's library: races
    python tools/device_view_race.py --trials 50                              # this tree: 0 failures
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = 0.016666


def load_package(tree):
    pkg = os.path.join(tree or ROOT, "webgl-fluid-simulation_amd")
    sys.path.insert(0, pkg)
    import fluid_hip
    assert os.path.dirname(os.path.dirname(os.path.abspath(fluid_hip.__file__))) == os.path.abspath(pkg), fluid_hip.__file__
    return fluid_hip


def raw_view(sim, name):
    """the field through a bare fluid_field_device_ptr: no ordering call, whatever the package's device_view does"""
    import torch
    from fluid_hip import _abi
    ptr = C.c_void_p()
    sim._check(sim._lib.fluid_field_device_ptr(sim._ctx, _abi.FIELD_IDS[name], C.byref(ptr)))
    fi = sim._info(name)
    shape = (fi.rows + 2 * fi.halo, fi.pitch, fi.channels)

    class _DeviceArray:
        __cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr.value, False), "version": 2}

    return torch.as_tensor(_DeviceArray(), device="cuda:0")[:, :fi.width]


def make_pair(fluid_hip, size, iters):
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    sims = [fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=0, schedule=s, random=fluid_hip.mulberry32(1234)) for s in ("passes", "fused")]
    for s in sims:
        s.multipleSplats(20)
    return sims


def trial(fluid_hip, sims, mode, steps, field="dye"):
    """one splat + `steps` steps on both contexts, then the fused context's field the `mode` way against the per-pass context's host read.
    >= 16 steps: a packed dye that is unpacked after fewer advections stays RGBA for the next 256 (fluid_ctx::pack_holdoff) and there
    would be nothing to convert in the following trial."""
    import torch
    ref, fused = sims
    ref.multipleSplats(1)
    ref.step(DT, steps)
    want = torch.from_numpy(ref.read(field)).to("cuda:0")   # synchronous host path of the per-pass context
    torch.cuda.synchronize()
    fused.multipleSplats(1)                                 # (each context draws from its own mulberry32 stream: the same splat)
    fused.step(DT, steps)                                   # returns at once: the steps are queued on the context's stream
    if mode == "legacy":
        fused.sync()
        got = raw_view(fused, field)
    else:
        got = fused.device_view(field)
    eq = bool(torch.equal(got, want))          # enqueued at once on torch's current stream
    n_diff = 0 if eq else int((got != want).sum().item())
    torch.cuda.synchronize()
    fused.sync()
    late = bool(torch.equal(raw_view(fused, field), want))   # the same memory once everything has finished: the kernels are right either way
    return eq, n_diff, late


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", default=None, help="another checkout to load the package and its library from (default: this one)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--trials", type=int, default=50)
    ap.add_argument("--modes", default="legacy,unsynced,unsynced_velocity")
    args = ap.parse_args()
    fluid_hip = load_package(args.tree)
    from fluid_hip import _abi
    out = {"tree": args.tree or ".", "library": _abi.LIB_PATH, "abi": _abi.lib().fluid_abi_version(), "size": args.size, "iters": args.iters}
    sims = make_pair(fluid_hip, args.size, args.iters)
    for mode in args.modes.split(","):
        steps = 16 if mode == "legacy" else 100    # unsynced: ~50 ms of fused steps still queued when the view is taken
        fails, diffs, late_ok, t0 = 0, [], 0, time.time()
        for _ in range(args.trials):
            eq, n_diff, late = trial(fluid_hip, sims, mode, steps, field="velocity" if mode.endswith("velocity") else "dye")
            fails += 0 if eq else 1
            late_ok += 1 if late else 0
            if not eq:
                diffs.append(n_diff)
        out[mode] = {"trials": args.trials, "steps_per_trial": steps, "mismatches": fails, "differing_values": diffs[:10],
                     "equal_once_everything_finished": late_ok, "seconds": round(time.time() - t0, 1)}
    for s in sims:
        s.close()
    print(json.dumps(out))
    return 1 if any(out[m]["mismatches"] for m in args.modes.split(",")) else 0


if __name__ == "__main__":
    sys.exit(main())
