"""fluid_set_curl_output (ABI 10): the reference writes its curl texture in every step and reads it in the same step only (curlProgram /
vorticityProgram, script.js:1234-1243) — nothing outside step() looks at it.  A host that does not either switches the OUTPUT off: no step
stores the field (4 B/texel less per call of one step, the page's update() pattern, script.js:1176-1186).  Held here: every other field keeps
its bits, a read of the curl field that was not stored is an error (never an older step's field), and switching the output on again gives
the reference's curl back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 0.016666


@pytest.mark.parametrize("w,h,iters", [(4096, 4096, 50), (1024, 1024, 20), (520, 300, 12)])
@pytest.mark.parametrize("schedule", ["fused", "passes"])
def test_steps_without_the_curl_output_leave_every_other_field_the_same(w, h, iters, schedule):
    import fluid_hip
    if schedule == "passes" and w == 4096:
        pytest.skip("one size is enough for the schedule that always runs the curl pass")
    cfg = {"SIM_RESOLUTION": min(w, h), "DYE_RESOLUTION": min(w, h), "PRESSURE_ITERATIONS": iters}
    sims = [fluid_hip.FluidSim(canvas=(w, h), config=cfg, schedule=schedule, random=fluid_hip.mulberry32(11)) for _ in range(2)]
    try:
        sims[1].set_curl_output(False)
        for s in sims:
            s.multipleSplats(4)
            for _ in range(3):
                s.step(DT, 1)          # one step per call: update()'s pattern
            s.step(DT, 2)
        for k in ("velocity", "pressure", "divergence", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
        info = sims[1].schedule_info(1, DT)
        plain_fused = schedule == "fused" and not info["chained"] and not info["runs_ahead"]
        if plain_fused:
            assert info["curl_stores"] == 0, info
            with pytest.raises(fluid_hip.FluidError) as e:
                sims[1].read("curl")
            assert "curl" in str(e.value)
        else:   # the per-pass schedule and the small-grid launches that carry the next step's stencil stages write it anyway
            assert np.array_equal(sims[0].read("curl"), sims[1].read("curl"))
        sims[1].set_curl_output(True)
        for s in sims:
            s.step(DT, 1)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
    finally:
        for s in sims:
            s.close()
