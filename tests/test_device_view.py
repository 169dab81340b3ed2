"""The zero-copy boundary: memory handed out by fluid_field_device_ptr / FluidSim.device_view is what the reference's "DoubleFBO.read after
a call is the result" (script.js:1079-1106) promises — the finished field — under the ordering rule written in include/fluid_hip.h.

Round 4's driver bench was an error record (BENCH_r04.json: fused_vs_passes_4096 MISMATCH): the packed dye's conversion back to RGBA was
enqueued by the pointer query itself, behind the caller's sync, on a stream nothing else is ordered against.  No test read a raw pointer;
these do, at the headline size, with the sequences of tools/device_view_race.py (which shows the same trials FAILING on round 4's library:
profiles/r05/device_view_race.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    import fluid_hip
    import device_view_race as R
    sims = R.make_pair(fluid_hip, 4096, 50)
    yield fluid_hip, R, sims
    for s in sims:
        s.close()


def test_sync_then_raw_pointer_is_finished_memory(pair):
    """rule (1): `fluid_sync(); fluid_field_device_ptr();` — the conversion the query triggers is waited for inside the call (50 / 50)"""
    fluid_hip, R, sims = pair
    info = sims[1].schedule_info(16, R.DT)
    assert info["dye_packed"], "the fused 4096^2 context must be running the packed dye for this test to mean anything"
    for t in range(50):
        eq, n_diff, late = R.trial(fluid_hip, sims, "legacy", 16)
        assert late, "trial %d: the fused schedule's dye differs from the per-pass schedule's even after everything finished" % t
        assert eq, "trial %d: %d values read through the raw pointer differ (read while the conversion was writing them?)" % (t, n_diff)


@pytest.mark.parametrize("field", ["dye", "velocity"])
def test_device_view_orders_torch_behind_queued_steps(pair, field):
    """rule (2): 100 fused steps (~50 ms) still queued, no host sync, torch reads the view at once on its own stream"""
    fluid_hip, R, sims = pair
    for t in range(6):
        eq, n_diff, late = R.trial(fluid_hip, sims, "unsynced", 100, field=field)
        assert late and eq, "trial %d (%s): %d values differ" % (t, field, n_diff)


def test_context_waits_for_a_consumer_stream(pair):
    """the reverse hazard: a torch kernel still reading the view when the next step overwrites the buffer — wait_for_torch orders the step
    behind it.  A long torch reduction over the velocity view, then a step at once: the reduction must have seen the pre-step field."""
    import torch
    fluid_hip, R, sims = pair
    fused = sims[1]
    fused.sync()
    v = fused.device_view("velocity")
    want = v.clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        acc = torch.zeros_like(want)
        for _ in range(40):          # ~40 x 134 MB of reads on a side stream: still running when the step below is enqueued
            acc += v
        fused.wait_for_torch()       # current stream = side
    fused.step(R.DT, 4)              # swaps and overwrites the velocity buffers
    fused.sync()
    torch.cuda.synchronize()
    ref = torch.zeros_like(want)
    for _ in range(40):
        ref += want
    assert torch.equal(acc, ref)


def test_parity_in_run_of_the_bench_at_the_headline_size():
    """bench.py's own checker, the function the driver's run died in, at 4096^2 / 50: green, and it says what it compared per field"""
    import fluid_hip
    sys.path.insert(0, ROOT)
    import bench
    for _ in range(3):
        p = bench.parity_in_run(fluid_hip, 4096, 50, 0, "f32", with_oracle=False)
        assert p["ok"] and "bitwise" in p["fused_vs_passes_4096"], p
        assert all(f["equal"] for f in p["fields_4096"].values()) and set(p["fields_4096"]) == {"velocity", "pressure", "divergence", "curl", "dye"}
