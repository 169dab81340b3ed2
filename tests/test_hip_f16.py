"""fp16-STORAGE mode of the HIP path against the oracle's fp16 mode (the fp32 restatement with an fp16
round trip after every pass output — the parity target the survey names), through the C ABI.

Bar: BITWISE, everywhere.  Both sides restate the same fp32 arithmetic in the same order, sqrt and divide are correctly rounded on
both, and exp() is the same polynomial (the reference rasteriser's, fluid_math.h exp_reference / fluid_oracle.c fo_exp_reference), so
the fp16 values stored after every pass are identical — single passes, splats and multi-step runs with CURL = 30 alike."""
import numpy as np
import pytest

import scenario as S

pytestmark = pytest.mark.gpu

SIZES = [(37, 53), (64, 64), (250, 130), (512, 300)]


def half(a):
    with np.errstate(over="ignore"):
        return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def ulp16(x):
    """spacing of the fp16 grid at |x| (the smallest subnormal step near zero)"""
    return np.spacing(np.abs(x).astype(np.float16)).astype(np.float32)


def make(W, H, schedule="fused", config=None, storage="f16", seed=None):
    import fluid_hip
    cfg = {"SIM_RESOLUTION": min(W, H), "DYE_RESOLUTION": min(W, H)}
    cfg.update(config or {})
    sim = fluid_hip.FluidSim(canvas=(W, H), config=cfg, schedule=schedule, storage=storage,
                             random=fluid_hip.mulberry32(seed) if seed is not None else None)
    assert (sim.velocity.width, sim.velocity.height) == (W, H)
    return sim


def rand_state(W, H, seed):
    rng = np.random.default_rng(seed)
    return {"velocity": half(rng.normal(0, 80, (H, W, 2))), "pressure": half(rng.normal(0, 30, (H, W))),
            "divergence": half(rng.normal(0, 30, (H, W))), "curl": half(rng.normal(0, 30, (H, W))),
            "dye": half(np.abs(rng.normal(0, 1, (H, W, 4))))}


def load_state(sim, st):
    for k, v in st.items():
        sim.write(k, v)


def assert_same_halves(got, want, what):
    assert np.array_equal(got, half(got)), what                       # what is stored IS a half
    assert np.array_equal(got, want), (what, int((got != want).sum()))


def test_write_read_round_trip_rounds_to_nearest_even():
    rng = np.random.default_rng(1)
    W, H = 96, 40
    with make(W, H) as sim:
        for name, shape in (("velocity", (H, W, 2)), ("pressure", (H, W)), ("dye", (H, W, 4))):
            a = (rng.normal(0, 50, shape) * rng.choice([1e-6, 1e-3, 1.0, 100.0, 3000.0], shape)).astype(np.float32)
            sim.write(name, a)
            assert np.array_equal(sim.read(name), half(a)), name          # overflow -> inf, subnormals kept, ties to even
        info = sim._info("dye")
        assert info.bytes_per_channel == 2 and info.channels == 4


@pytest.mark.parametrize("W,H", SIZES)
def test_bitwise_passes(oracle, W, H):
    st = rand_state(W, H, 100 + W)
    R = oracle.round_half
    with make(W, H, "passes") as sim:
        load_state(sim, st)
        sim.run_pass("curl")
        assert np.array_equal(sim.read("curl"), R(oracle.curl(st["velocity"])))
        load_state(sim, st)
        sim.run_pass("divergence")
        assert np.array_equal(sim.read("divergence"), R(oracle.divergence(st["velocity"])))
        sim.run_pass("clear")
        assert np.array_equal(sim.read("pressure"), R(oracle.clear(st["pressure"], np.float32(0.8))))
        load_state(sim, st)
        sim.run_pass("gradsub")
        assert np.array_equal(sim.read("velocity"), R(oracle.gradsub(st["pressure"], st["velocity"])))


@pytest.mark.parametrize("schedule", ["passes", "fused"])
@pytest.mark.parametrize("iters", [1, 2, 7, 10, 11, 20, 50])
@pytest.mark.parametrize("W,H", [(64, 64), (250, 130), (256, 64), (512, 300), (1000, 40)])
def test_jacobi_bitwise_with_rounding_after_every_iteration(oracle, W, H, iters, schedule):
    st = rand_state(W, H, 7 * W + iters)
    with make(W, H, schedule) as sim:
        load_state(sim, st)
        sim.run_pass("jacobi", iters=iters)
        got = sim.read("pressure")
    p = st["pressure"]
    for _ in range(iters):
        p = oracle.round_half(oracle.jacobi(p, st["divergence"]))      # the reference renders each iteration into a half-float texture
    assert np.array_equal(got, p)


@pytest.mark.parametrize("W,H", SIZES)
def test_sqrt_divide_passes_bitwise(oracle, W, H):
    st = rand_state(W, H, 300 + H)
    R, dt = oracle.round_half, np.float32(0.016666)
    with make(W, H, "passes") as sim:
        P = sim.params()
        load_state(sim, st)
        sim.run_pass("vorticity")
        assert_same_halves(sim.read("velocity"), R(oracle.vorticity(st["velocity"], st["curl"], P.curl, dt)), "vorticity")
        small = dict(st, velocity=half(st["velocity"] * 0.05))             # back-traces of a few texels
        load_state(sim, small)
        sim.run_pass("advect_velocity")
        assert_same_halves(sim.read("velocity"), R(oracle.advect(small["velocity"], small["velocity"], dt, P.velocity_dissipation)), "advect velocity")
        load_state(sim, small)
        sim.run_pass("advect_dye")
        assert_same_halves(sim.read("dye"), R(oracle.advect(small["velocity"], small["dye"], dt, P.density_dissipation)), "advect dye")


@pytest.mark.parametrize("curl", [0, 30])
@pytest.mark.parametrize("canvas,cfg", [((256, 256), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}),
                                        ((600, 300), {"SIM_RESOLUTION": 48, "DYE_RESOLUTION": 160})])
def test_steps_against_the_oracle(oracle, canvas, cfg, curl):
    cfg = dict(cfg, CURL=curl, PRESSURE_ITERATIONS=20)
    ref = oracle.RefSim(canvas=canvas, config=cfg, seed=11, storage="f16")
    import fluid_hip
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, storage="f16", random=fluid_hip.mulberry32(11)) as sim:
        ref.multiple_splats(4); sim.multipleSplats(4)
        for k in ("velocity", "dye"):
            assert_same_halves(sim.read(k), ref.fields()[k], "splat " + k)
        ref.step(0.016666, 3); sim.step(0.016666, 3)
        got = sim.fields()
    want = ref.fields()
    for k in S.FIELDS:
        assert got[k].shape == want[k].shape
        assert np.array_equal(got[k], want[k]), (k, S.rel_err(got[k], want[k]))


@pytest.mark.parametrize("canvas,cfg", [((512, 512), {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 128, "PRESSURE_ITERATIONS": 50}),
                                        ((512, 256), {"SIM_RESOLUTION": 100, "DYE_RESOLUTION": 260, "PRESSURE_ITERATIONS": 23}),
                                        ((1000, 500), {"SIM_RESOLUTION": 250, "DYE_RESOLUTION": 250, "PRESSURE_ITERATIONS": 31}),   # W % 4 != 0 on neither grid… 500 x 250
                                        ((2048, 2048), {"SIM_RESOLUTION": 2048, "DYE_RESOLUTION": 2048, "PRESSURE_ITERATIONS": 50})])
def test_fused_equals_passes_bitwise(canvas, cfg):
    out = []
    for schedule in ("passes", "fused"):
        import fluid_hip
        with fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=schedule, storage="f16", random=fluid_hip.mulberry32(5)) as sim:
            sim.multipleSplats(6)
            sim.step(0.016666, 4)
            out.append(sim.fields())
    for k in S.FIELDS:
        assert np.array_equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("ty,tx", [(2, 1), (4, 1), (2, 2)])
def test_stripe_and_tile_groups_equal_the_single_domain_bitwise(ty, tx):
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 30}
    with fluid_hip.FluidSim(canvas=(512, 512), config=cfg, storage="f16", random=fluid_hip.mulberry32(9)) as sim:
        sim.multipleSplats(5)
        sim.step(0.016666, 3)
        want = sim.fields()
    g = StripeGroup(ty * tx, canvas=(512, 512), config=cfg, halo=24, random=fluid_hip.mulberry32(9), tiles_x=tx, storage="f16")
    try:
        g.multipleSplats(5)
        g.step(0.016666, 3)
        g.check_halo()
        for k in S.FIELDS:
            assert np.array_equal(g.read(k), want[k]), k
        assert g.exchanges > 0
    finally:
        g.close()


def test_resize_resamples_and_rounds(oracle):
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}
    ref = oracle.RefSim(canvas=(512, 512), config=cfg, seed=3, storage="f16")
    import fluid_hip
    with fluid_hip.FluidSim(canvas=(512, 512), config=cfg, storage="f16", random=fluid_hip.mulberry32(3)) as sim:
        ref.multiple_splats(4); sim.multipleSplats(4)
        for s in (ref, sim):
            s.config.update({"SIM_RESOLUTION": 96, "DYE_RESOLUTION": 200})
        st = {k: sim.read(k) for k in ("velocity", "dye")}                    # resample the HIP state with the oracle: isolates the pass
        ref.vel[0], ref.dye[0] = st["velocity"], st["dye"]
        ref.init_framebuffers(); sim.initFramebuffers()
        got = sim.fields()
    want = ref.fields()
    assert got["velocity"].shape == (96, 96, 2) and got["dye"].shape == (200, 200, 4)
    for k in ("velocity", "dye"):
        assert_same_halves(got[k], want[k], "resize " + k)
    for k in ("pressure", "divergence", "curl"):
        assert not got[k].any()


def test_display_of_an_fp16_dye_field():
    """the compositor widens dye.read (exact) and works in fp32: same frame as the numpy restatement on the same texels"""
    import fluid_hip
    from oracle import display as D
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 128}
    with fluid_hip.FluidSim(canvas=(256, 256), config=cfg, storage="f16", random=fluid_hip.mulberry32(2)) as sim:
        sim.multipleSplats(5)
        sim.step(0.016666, 2)
        dye = sim.read("dye")
        w, h = D.get_resolution(D.DISPLAY_DEFAULTS["CAPTURE_RESOLUTION"], 256, 256)
        frame = sim.render(w, h)
        assert np.array_equal(sim.read("dye"), dye)                            # rendering leaves dye.read alone
    want = D.capture(dye, (256, 256), dict(D.DISPLAY_DEFAULTS), None)
    err = np.abs(frame - want["frame"]).max() / max(float(np.abs(want["frame"]).max()), 1e-30)
    assert err <= 4e-6, err


@pytest.mark.parametrize("world,tx,halo,cfg", [(2, 1, 56, {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}),
                                               (4, 2, 24, {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 25})])
def test_rccl_exchange_of_half_fields_with_several_ranks_bitwise(world, tx, halo, cfg):
    """the native RCCL driver moves BYTES (ncclChar): rank threads against the in-process RCCL stand-in, fp16 fields,
    interior-first overlap on the fused half kernels; assembled result bitwise equal to the single fp16 domain"""
    import json
    import os
    import subprocess
    import sys
    from test_stripes_gpu import fake_rccl_lib
    here = os.path.dirname(os.path.abspath(__file__))
    lib = fake_rccl_lib()
    args = {"world": world, "tiles_x": tx, "halo": halo, "config": cfg, "canvas": [512, 512], "steps": 2, "storage": "f16"}
    r = subprocess.run([sys.executable, os.path.join(here, "fake_rccl", "run_ranks.py"), json.dumps(args)], env=dict(os.environ, FLUID_RCCL_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["exchanges"] > 0, out


def test_node_host_in_fp16_mode_agrees_with_python_bitwise(tmp_path):
    """createFluid({storage: 'f16'}) — the JavaScript host reaches the same half fields as the Python host"""
    import shutil
    import fluid_hip
    if shutil.which("node") is None:
        pytest.skip("node not installed")
    from test_node_shim import node
    cfg = {"SIM_RESOLUTION": 96, "DYE_RESOLUTION": 160, "PRESSURE_ITERATIONS": 25}
    args = {"canvas": {"width": 600, "height": 300}, "config": cfg, "seed": 77, "randomSplats": 5, "steps": 3, "dt": 0.016666,
            "schedule": "fused", "storage": "f16", "out": str(tmp_path / "fields.bin")}
    node("run_scenario.js", args)
    with fluid_hip.FluidSim(canvas=(600, 300), config=cfg, storage="f16", random=fluid_hip.mulberry32(77)) as sim:
        sim.multipleSplats(5)
        for _ in range(3):
            sim.step(0.016666)
        want = sim.fields()
    raw = np.fromfile(args["out"], dtype=np.float32)
    off = 0
    for k in S.FIELDS:
        n = want[k].size
        got = raw[off:off + n].reshape(want[k].shape)
        assert np.array_equal(got, want[k]) and np.array_equal(got, half(got)), k
        off += n
