"""The chained pressure loop's failure path and its behaviour under a foreign load (script.js:1259-1266 is the loop).

k_jacobi_tb_chain lets a tile of block l wait, inside one launch, for the tile rows of block l - 1 around it.  The wait is bounded in
wall-clock time; a workgroup that gives up raises a word of mapped host memory and computes on stale data rather than hang the device.
What must then hold, and is held here on the GPU with the give-up path FORCED (lab build: one tile never counts itself, 20 ms bound):
  * the error is the error of whichever call returns data or a status first — fluid_sync, a field read, a rendered frame's readback, a raw
    device pointer, fluid_stream_wait_context once the device has run the launch — never a silently wrong field or frame;
  * the context recovers by itself: it keeps to one launch per block of iterations from then on and, from the fields as they are, steps
    bit for bit like the one-kernel-per-pass schedule;
and, with nothing forced (product build): chained steps taken while another stream keeps every CU busy with a foreign kernel leave exactly
the bits of an undisturbed run — a dependency that shows up late is waited for, not guessed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip_probes.so")
DT = 0.016666

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "webgl-fluid-simulation_amd"))
import numpy as np
import fluid_hip
from fluid_hip import _abi
import ctypes as C
exit_name = %(exit)r
cfg = {"SIM_RESOLUTION": 4096, "DYE_RESOLUTION": 4096, "PRESSURE_ITERATIONS": 50}
out = {"flavor": fluid_hip.lib().fluid_build_flavor().decode() if hasattr(fluid_hip.lib(), "fluid_build_flavor") else "?"}
sim = fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, schedule="fused", random=fluid_hip.mulberry32(5))
ref = fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, schedule="passes", random=fluid_hip.mulberry32(5))
out["chained_before"] = bool(sim.schedule_info(1, %(dt)r)["jacobi_chained"])
sim.multipleSplats(4)
ref.multipleSplats(4)        # (keeps the two random streams in step; its fields are overwritten below)
sim.step(%(dt)r, 1)          # enqueued: the launch gives up on the device some 20 ms from now
err = None
try:
    if exit_name == "sync":
        sim.sync()
    elif exit_name == "read":
        sim.read("pressure")
    elif exit_name == "frame":
        sim.render(256, 256)
    elif exit_name == "device_ptr":
        import torch
        torch.cuda.synchronize()          # the caller synchronised with the device by other means ...
        sim.device_view("velocity")       # ... and asks for a raw pointer: no pointer to fields that are known to be wrong
    elif exit_name == "stream_wait":
        import torch
        torch.cuda.synchronize()
        with torch.cuda.device(0):
            s = torch.cuda.current_stream()
            sim._check(sim._lib.fluid_stream_wait_context(sim._ctx, C.c_void_p(s.cuda_stream)))
except fluid_hip.FluidError as ex:
    err = str(ex)
out["error"] = err
# the context goes on by itself: one launch per block from here on, and from the fields AS THEY ARE it steps like the per-pass schedule
out["chained_after"] = bool(sim.schedule_info(1, %(dt)r)["jacobi_chained"])
sim.sync()                                # (the error was reported once: this call is clean)
for k in ("velocity", "pressure", "divergence", "curl", "dye"):
    ref.write(k, np.nan_to_num(sim.read(k), nan=0.0, posinf=0.0, neginf=0.0))
    sim.write(k, ref.read(k))
for s in (sim, ref):
    s.multipleSplats(2)
    s.step(%(dt)r, 2)
same = {k: bool(np.array_equal(sim.read(k).view(np.uint32), ref.read(k).view(np.uint32))) for k in ("velocity", "pressure", "divergence", "curl", "dye")}   # (bit patterns: the fields grew from garbage and may hold NaNs)
out["fields_equal"] = same
out["equal_after"] = all(same.values())
print(json.dumps(out))
'''


@pytest.mark.parametrize("exit_name", ["sync", "read", "frame", "device_ptr", "stream_wait"])
def test_a_chained_loop_that_gives_up_is_an_error_at_every_exit_and_the_context_recovers(exit_name):
    if not os.path.exists(LAB):
        pytest.fail("libfluid_hip_probes.so is not built (make PROBES=1): the give-up path needs the lab build's withheld counter")
    env = dict(os.environ, FLUID_HIP_LIB=LAB, FLUID_CHAIN_WITHHOLD="7", FLUID_CHAIN_TIMEOUT_MS="20")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "exit": exit_name, "dt": DT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["chained_before"] is True, out
    assert out["error"] is not None and "gave up" in out["error"], out          # the launch's failure surfaced at THIS exit
    assert out["chained_after"] is False, out                                   # ... the context keeps to plain launches from then on
    assert out["equal_after"] is True, out                                      # ... and computes the right thing again


def test_chained_steps_under_a_foreign_load_leave_the_same_bits():
    """another stream keeps the chip busy with large matrix products (every CU, tens of milliseconds) while the chained steps run"""
    import torch
    import fluid_hip
    cfg = {"SIM_RESOLUTION": 4096, "DYE_RESOLUTION": 4096, "PRESSURE_ITERATIONS": 50}
    sims = [fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, schedule=s, random=fluid_hip.mulberry32(17)) for s in ("passes", "fused")]
    try:
        assert sims[1].schedule_info(1, DT)["jacobi_chained"]
        for s in sims:
            s.multipleSplats(6)
        sims[0].step(DT, 6)
        sims[0].sync()
        side = torch.cuda.Stream()
        a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
        b = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
        torch.cuda.synchronize()
        for burst in range(3):          # uneven load: bursts of foreign work start while steps are in flight, and end while others are
            with torch.cuda.stream(side):
                for _ in range(6):
                    c = a @ b           # ~1.1 TFLOP each: milliseconds of every CU
            sims[1].step(DT, 2)
        sims[1].sync()                  # (raises if a tile gave up waiting)
        torch.cuda.synchronize()
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
        del c
    finally:
        for s in sims:
            s.close()
