"""Decomposition invariance on ONE GPU: N stripe contexts of libfluid_hip.so on the same device (one thread
each, LocalComm mailboxes in place of RCCL) must reproduce the single-domain HIP result BITWISE — the same
windowed kernels, ghost-row staging and exchange schedule that bench.py --gpus N runs over RCCL."""
import numpy as np
import pytest

import scenario as S

pytestmark = pytest.mark.gpu

CASES = [
    # canvas, config, halo, world, steps, schedule
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 50}, 32, 2, 2, "fused"),
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 50}, 32, 4, 2, "passes"),
    ((512, 512), {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 20}, 16, 4, 2, "fused"),
    ((256, 1024), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 30}, 12, 8, 1, "fused"),
    ((512, 512), {"SIM_RESOLUTION": 1024, "DYE_RESOLUTION": 1024, "PRESSURE_ITERATIONS": 50}, 32, 4, 1, "fused"),
]


@pytest.mark.parametrize("canvas,cfg,halo,world,steps,schedule", CASES)
def test_hip_stripes_equal_single_domain_bitwise(canvas, cfg, halo, world, steps, schedule):
    import fluid_hip
    from fluid_hip.stripes import run_local_stripes
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=schedule, random=fluid_hip.mulberry32(9)) as one:
        one.multipleSplats(6)
        for _ in range(steps):
            one.step(0.016666)
        want = one.fields()

    def body(sim):
        sim.random = fluid_hip.mulberry32(9)
        sim.multipleSplats(6)
        for _ in range(steps):
            sim.step(0.016666)
        sim.sync()
        sim.check_halo()
        return {k: sim.read_local(k) for k in S.FIELDS}

    res = run_local_stripes(world, body, canvas=canvas, config=cfg, halo=halo, schedule=schedule, device=0)
    for k in S.FIELDS:
        got = np.concatenate([r[k] for r in res], axis=0)
        assert got.shape == want[k].shape
        assert np.array_equal(got, want[k]), k


def test_hip_halo_overflow_raises():
    import fluid_hip
    from fluid_hip.stripes import run_local_stripes
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 4}

    def body(sim):
        sim.splat(0.5, 0.5, 0.0, 90000.0, (1, 1, 1))
        sim.step(0.016666)
        sim.check_halo()

    with pytest.raises(fluid_hip.FluidError) as e:
        run_local_stripes(2, body, canvas=(256, 256), config=cfg, halo=4, device=0)
    assert e.value.status == -5
