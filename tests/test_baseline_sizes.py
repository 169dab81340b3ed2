"""BASELINE.json configs[3] and configs[4] at FULL size on ONE MI355X (288 GB of HBM hold both grids): the CPU oracle is
far too slow to be the checker there, so the checks are size-independent properties — schedule equivalence (the fused
kernels reproduce the reference pass structure bit for bit) and decomposition invariance (the stripe set the 4- / 8-GPU
runs use, stepped by the native plan of csrc/fluid_stripes.cpp inside one process, reproduces the single domain bit for
bit).  Fields are compared one at a time to bound host memory."""
import gc

import numpy as np
import pytest

import scenario as S

pytestmark = pytest.mark.gpu


def _free():
    gc.collect()


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_config3_8192_sq_fused_equals_passes_bitwise(storage):
    """configs[3] workload: 8192^2 sim = dye, 50 Jacobi iterations (fp32 fields: the headline; and the fp16-storage side mode)"""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": 8192, "DYE_RESOLUTION": 8192, "PRESSURE_ITERATIONS": 50}
    sims = [fluid_hip.FluidSim(canvas=(8192, 8192), config=cfg, schedule=s, random=fluid_hip.mulberry32(8), storage=storage)
            for s in ("passes", "fused")]
    try:
        for s in sims:
            s.multipleSplats(6)
            s.step(0.016666, 1)
        for k in S.FIELDS:
            a, b = sims[0].read(k), sims[1].read(k)
            assert np.array_equal(a, b), k
            del a, b
            _free()
    finally:
        for s in sims:
            s.close()


def test_config3_8192_sq_four_stripes_equal_single_domain_bitwise():
    """configs[3] on 4 GPUs as four row stripes of 8192 x 2048 (DESIGN.md §7 explains why 1-D stripes and not 2 x 2)"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 8192, "DYE_RESOLUTION": 8192, "PRESSURE_ITERATIONS": 50}
    one = fluid_hip.FluidSim(canvas=(8192, 8192), config=cfg, random=fluid_hip.mulberry32(21))
    g = StripeGroup(4, canvas=(8192, 8192), config=cfg, halo=56, random=fluid_hip.mulberry32(21))
    try:
        one.multipleSplats(6); g.multipleSplats(6)
        one.step(0.016666, 2); g.step(0.016666, 2)
        g.check_halo()
        assert g.exchanges == 2 * 2                      # halo 56: {velocity, pressure} and {velocity, dye} per step
        for k in S.FIELDS:
            a, b = one.read(k), g.read(k)
            assert np.array_equal(a, b), k
            del a, b
            _free()
    finally:
        one.close(); g.close()


def test_config4_16384_sq_200_iterations_eight_stripes_equal_single_domain_bitwise():
    """configs[4]: 16384^2, 200 Jacobi iterations, eight stripes of 16384 x 2048 — the 8-GPU layout, exact sizes"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 16384, "DYE_RESOLUTION": 16384, "PRESSURE_ITERATIONS": 200}
    one = fluid_hip.FluidSim(canvas=(16384, 16384), config=cfg, random=fluid_hip.mulberry32(4))
    g = StripeGroup(8, canvas=(16384, 16384), config=cfg, halo=56, random=fluid_hip.mulberry32(4))
    try:
        one.multipleSplats(5); g.multipleSplats(5)
        one.step(0.016666, 1); g.step(0.016666, 1)
        g.check_halo()
        plan = fluid_hip._abi.stripe_plan(56, 56, 200, 20, 20)
        assert g.exchanges == sum(1 for op in plan if op[0] == "exchange") == 5     # 200 iterations = 4 blocks of 50 (whole launches: round 6; 53 + 53 + 53 + 41 before)
        for k in ("pressure", "divergence", "curl", "velocity", "dye"):
            a, b = one.read(k), g.read(k)
            assert np.array_equal(a, b), k
            del a, b
            _free()
    finally:
        one.close(); g.close()


def test_config3_8192_sq_two_by_two_tiles_equal_single_domain_bitwise():
    """configs[3] as BASELINE.json words it: 8192^2, 2 x 2 domain decomposition (four tiles of 4096 x 4096), 50 iterations"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 8192, "DYE_RESOLUTION": 8192, "PRESSURE_ITERATIONS": 50}
    one = fluid_hip.FluidSim(canvas=(8192, 8192), config=cfg, random=fluid_hip.mulberry32(21))
    g = StripeGroup(4, canvas=(8192, 8192), config=cfg, halo=56, random=fluid_hip.mulberry32(21), tiles_x=2)
    try:
        one.multipleSplats(6); g.multipleSplats(6)
        one.step(0.016666, 2); g.step(0.016666, 2)
        g.check_halo()
        assert [(e.info("velocity").col0, e.info("velocity").row0) for e in g.engines] == [(0, 0), (4096, 0), (0, 4096), (4096, 4096)]
        for k in S.FIELDS:
            a, b = one.read(k), g.read(k)
            assert np.array_equal(a, b), k
            del a, b
            _free()
    finally:
        one.close(); g.close()
