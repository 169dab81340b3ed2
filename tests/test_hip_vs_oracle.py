"""HIP kernels vs the CPU oracle on identical seeded inputs, through the C ABI.

Bar: BITWISE everywhere, at every size — the passes are add / multiply arithmetic in a fixed order (-ffp-contract=off on both
sides), sqrt and divide (vorticity, advection) are correctly rounded on both sides, and exp (splat) is the same fixed polynomial
(the reference rasteriser's, tests/tolerances.py); so whole steps with splats agree bit for bit as well."""
import numpy as np
import pytest

import scenario as S


pytestmark = pytest.mark.gpu

# (W, H): odd sizes, W % 4 != 0 (the last quad of a row is partly padding), tiles smaller/larger than a Jacobi tile
SIZES = [(37, 53), (64, 64), (250, 130), (256, 64), (512, 300), (1000, 40), (1024, 1024)]


def make_sim(W, H, schedule, config=None, dye=None):
    import fluid_hip
    cfg = {"SIM_RESOLUTION": min(W, H), "DYE_RESOLUTION": min(W, H)}
    cfg.update(config or {})
    sim = fluid_hip.FluidSim(canvas=(W, H), config=cfg, schedule=schedule)
    assert (sim.velocity.width, sim.velocity.height) == (W, H)
    return sim


def rand_state(W, H, seed, dW=None, dH=None):
    rng = np.random.default_rng(seed)
    dW, dH = dW or W, dH or H
    return {"velocity": rng.normal(0, 80, (H, W, 2)).astype(np.float32),
            "pressure": rng.normal(0, 30, (H, W)).astype(np.float32),
            "divergence": rng.normal(0, 30, (H, W)).astype(np.float32),
            "curl": rng.normal(0, 30, (H, W)).astype(np.float32),
            "dye": np.abs(rng.normal(0, 1, (dH, dW, 4))).astype(np.float32)}


def load_state(sim, st):
    for k, v in st.items():
        sim.write(k, v)


@pytest.mark.parametrize("W,H", SIZES)
def test_bitwise_passes(oracle, W, H):
    st = rand_state(W, H, 100 + W)
    sim = make_sim(W, H, "passes")
    try:
        load_state(sim, st)
        sim.run_pass("curl")
        assert np.array_equal(sim.read("curl"), oracle.curl(st["velocity"]))
        load_state(sim, st)
        sim.run_pass("divergence")
        assert np.array_equal(sim.read("divergence"), oracle.divergence(st["velocity"]))
        sim.run_pass("clear")
        assert np.array_equal(sim.read("pressure"), oracle.clear(st["pressure"], np.float32(0.8)))
        load_state(sim, st)
        sim.run_pass("gradsub")
        assert np.array_equal(sim.read("velocity"), oracle.gradsub(st["pressure"], st["velocity"]))
    finally:
        sim.close()


@pytest.mark.parametrize("schedule", ["passes", "fused"])
@pytest.mark.parametrize("iters", [1, 2, 7, 8, 9, 20, 50])
@pytest.mark.parametrize("W,H", [(64, 64), (250, 130), (256, 64), (512, 300), (1000, 40), (1024, 1024)])
def test_jacobi_bitwise(oracle, W, H, iters, schedule):
    if (W, H) == (1024, 1024) and iters not in (8, 50):
        pytest.skip("large case only at the interesting iteration counts")
    st = rand_state(W, H, 7 * W + iters)
    sim = make_sim(W, H, schedule)
    try:
        load_state(sim, st)
        sim.run_pass("jacobi", iters=iters)
        got = sim.read("pressure")
    finally:
        sim.close()
    p = st["pressure"]
    for _ in range(iters):
        p = oracle.jacobi(p, st["divergence"])
    assert np.array_equal(got, p)


@pytest.mark.parametrize("W,H", SIZES)
def test_ulp_passes(oracle, W, H):
    st = rand_state(W, H, 200 + H)
    dt = np.float32(0.016666)
    sim = make_sim(W, H, "passes")
    try:
        load_state(sim, st)
        sim.run_pass("vorticity")
        ref = oracle.vorticity(st["velocity"], st["curl"], np.float32(30), dt)
        assert np.array_equal(sim.read("velocity"), ref)
        load_state(sim, st)
        sim.run_pass("advect_velocity")
        ref = oracle.advect(st["velocity"], st["velocity"], dt, np.float32(0.2))
        assert np.array_equal(sim.read("velocity"), ref)
        load_state(sim, st)
        sim.run_pass("advect_dye")
        ref = oracle.advect(st["velocity"], st["dye"], dt, np.float32(1.0))
        assert np.array_equal(sim.read("dye"), ref)
    finally:
        sim.close()


def test_vorticity_clamp(oracle):
    W = H = 96
    st = rand_state(W, H, 5)
    st["velocity"] = (st["velocity"] * 12).astype(np.float32)  # sigma ~ 960: many texels beyond +-1000
    sim = make_sim(W, H, "passes")
    try:
        load_state(sim, st)
        sim.run_pass("vorticity")
        got = sim.read("velocity")
    finally:
        sim.close()
    assert np.abs(got).max() == 1000.0
    assert np.array_equal(got, oracle.vorticity(st["velocity"], st["curl"], np.float32(30), np.float32(0.016666)))


def test_advect_dye_cross_resolution(oracle):
    import fluid_hip
    sim = fluid_hip.FluidSim(canvas=(512, 512), config={"SIM_RESOLUTION": 48, "DYE_RESOLUTION": 160}, schedule="passes")
    st = rand_state(48, 48, 9, 160, 160)
    try:
        load_state(sim, st)
        sim.run_pass("advect_dye")
        got = sim.read("dye")
    finally:
        sim.close()
    ref = oracle.advect(st["velocity"], st["dye"], np.float32(0.016666), np.float32(1.0))
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("W,H", [(64, 64), (130, 50), (400, 200)])
def test_splat(oracle, W, H):
    sim = make_sim(W, H, "passes")
    st = rand_state(W, H, 33)
    try:
        load_state(sim, st)
        sim.splat(0.31, 0.72, 412.5, -230.25, {"r": 1.2, "g": 0.3, "b": 0.05})
        gv, gd = sim.read("velocity"), sim.read("dye")
    finally:
        sim.close()
    aspect = W / H
    radius = 0.25 / 100.0 * (aspect if aspect > 1 else 1)
    f = oracle.f32
    rv = oracle.splat(st["velocity"], f(0.31), f(0.72), f(aspect), f(radius), (f(412.5), f(-230.25), 0.0))
    rd = oracle.splat(st["dye"], f(0.31), f(0.72), f(aspect), f(radius), (f(1.2), f(0.3), f(0.05)))
    assert np.array_equal(gv, rv)
    assert np.array_equal(gd, rd)
    assert np.all(gd[..., 3] == 1.0)


@pytest.mark.parametrize("schedule", ["passes", "fused"])
@pytest.mark.parametrize("canvas,cfg,steps", [
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 50}, 1),
    ((1024, 512), {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 192, "PRESSURE_ITERATIONS": 20}, 2),
    ((512, 512), {"SIM_RESOLUTION": 1024, "DYE_RESOLUTION": 1024, "PRESSURE_ITERATIONS": 50}, 1),
    ((300, 500), {"SIM_RESOLUTION": 90, "DYE_RESOLUTION": 90, "PRESSURE_ITERATIONS": 13, "CURL": 0}, 5),
])
def test_full_step_vs_oracle(oracle, canvas, cfg, steps, schedule):
    import fluid_hip
    ref = oracle.RefSim(canvas=canvas, config=cfg, seed=42)
    sim = fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=schedule, random=fluid_hip.mulberry32(42))
    try:
        a = ref.multiple_splats(6)
        b = sim.multipleSplats(6)
        assert np.array_equal(np.array(a), np.array(b))
        ref.step(0.016666, steps)
        sim.step(0.016666, steps)
        got = sim.fields()
    finally:
        sim.close()
    want = ref.fields()
    for k in S.FIELDS:
        assert got[k].shape == want[k].shape
        assert np.array_equal(got[k], want[k]), (k, S.rel_err(got[k], want[k]))   # whole steps, splats included: the same bits


def test_resize_preserves_dye_and_velocity(oracle):
    import fluid_hip
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}
    ref = oracle.RefSim(canvas=(512, 512), config=cfg, seed=3)
    sim = fluid_hip.FluidSim(canvas=(512, 512), config=cfg, random=fluid_hip.mulberry32(3))
    try:
        ref.multiple_splats(4); sim.multipleSplats(4)
        ref.step(0.016666, 2); sim.step(0.016666, 2)
        for s in (ref, sim):
            s.config.update({"SIM_RESOLUTION": 96, "DYE_RESOLUTION": 200})
        ref.init_framebuffers(); sim.initFramebuffers()
        got = sim.fields()
        # a second initFramebuffers() with unchanged sizes keeps dye/velocity, re-zeroes the rest (script.js:1117-1118, 1004-1006)
        sim.step(0.016666, 1)
        v_before = sim.read("velocity")
        sim.initFramebuffers()
        assert np.array_equal(sim.read("velocity"), v_before)
        assert not sim.read("pressure").any() and not sim.read("curl").any() and not sim.read("divergence").any()
    finally:
        sim.close()
    want = ref.fields()
    assert got["velocity"].shape == (96, 96, 2) and got["dye"].shape == (200, 200, 4)
    for k in ("velocity", "dye"):
        assert np.array_equal(got[k], want[k]), k
    for k in ("pressure", "divergence", "curl"):
        assert not got[k].any()


def test_failed_resize_keeps_the_fields():
    """a resize that cannot be allocated reports out-of-memory and leaves the context as it was (still steps, same bits)"""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 128}
    sim = fluid_hip.FluidSim(canvas=(512, 512), config=cfg, random=fluid_hip.mulberry32(5))
    twin = fluid_hip.FluidSim(canvas=(512, 512), config=cfg, random=fluid_hip.mulberry32(5))
    try:
        sim.multipleSplats(3); twin.multipleSplats(3)
        sim.step(0.016666, 2); twin.step(0.016666, 2)
        rc = sim._lib.fluid_resize(sim._ctx, 1 << 19, 1 << 19, 1 << 19, 1 << 19)  # 2 TB per velocity buffer
        assert rc == fluid_hip._abi.ERR_OOM, rc
        assert b"hipMalloc" in sim._lib.fluid_last_error(sim._ctx)
        sim.step(0.016666, 2); twin.step(0.016666, 2)
        a, b = sim.fields(), twin.fields()
        for k in S.FIELDS:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    finally:
        sim.close(); twin.close()


def test_framebuffer_to_texture_padding():
    import fluid_hip
    sim = fluid_hip.FluidSim(canvas=(64, 64), config={"SIM_RESOLUTION": 8, "DYE_RESOLUTION": 8})
    try:
        sim.splat(0.5, 0.5, 10.0, -5.0, {"r": 1, "g": 2, "b": 3})
        v = sim.framebufferToTexture(sim.velocity.read).reshape(8, 8, 4)
        p = sim.framebufferToTexture(sim.pressure.read).reshape(8, 8, 4)
        d = sim.framebufferToTexture("dye").reshape(8, 8, 4)
    finally:
        sim.close()
    assert np.all(v[..., 2] == 0) and np.all(v[..., 3] == 1) and v[..., 0].max() > 0
    assert np.all(p[..., 0] == 0) and np.all(p[..., 3] == 1)
    assert np.all(d[..., 3] == 1)


@pytest.mark.parametrize("W,H", [(64, 64), (256, 64), (512, 300), (1000, 40), (2048, 70), (1024, 1024)])
def test_fused_curl_vorticity_divergence(oracle, W, H):
    """the fused K1+K2+K3 kernel: curl bitwise; velocity to libm ulps; divergence bitwise GIVEN the kernel's own velocity"""
    st = rand_state(W, H, 300 + W)
    dt = np.float32(0.016666)
    sim = make_sim(W, H, "fused")
    try:
        load_state(sim, st)
        sim.run_pass("curl_vorticity_divergence")
        crl, vel, div = sim.read("curl"), sim.read("velocity"), sim.read("divergence")
    finally:
        sim.close()
    want_curl = oracle.curl(st["velocity"])
    assert np.array_equal(crl, want_curl)
    assert np.array_equal(vel, oracle.vorticity(st["velocity"], want_curl, np.float32(30), dt))
    assert np.array_equal(div, oracle.divergence(vel))


@pytest.mark.parametrize("W,H", [(64, 64), (250, 130), (512, 300), (1024, 1024)])
def test_fused_advect(oracle, W, H):
    st = rand_state(W, H, 400 + W)
    dt = np.float32(0.016666)
    sim = make_sim(W, H, "fused")
    try:
        load_state(sim, st)
        sim.run_pass("advect")
        vel, dye = sim.read("velocity"), sim.read("dye")
    finally:
        sim.close()
    assert np.array_equal(vel, oracle.advect(st["velocity"], st["velocity"], dt, np.float32(0.2)))
    # the dye back-trace uses the kernel's own new velocity: feed that to the oracle
    assert np.array_equal(dye, oracle.advect(vel, st["dye"], dt, np.float32(1.0)))
