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
    # a width that is not a multiple of 4 (250 x 128 with a 500 x 256 dye grid): the ghost rows travel with their padding columns (pitch 252)
    ((250, 128), {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 23}, 16, 2, 2, "fused"),
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


# ---- the NATIVE driver: libfluid_hip.so runs the plan itself (csrc/fluid_stripes.cpp) ------------------------------
GROUP_CASES = CASES + [
    ((512, 512), {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}, 56, 2, 2, "fused"),   # 2 exchanges per step
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 0, "CURL": 0}, 8, 4, 2, "fused"),
]


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("canvas,cfg,halo,world,steps,schedule", GROUP_CASES)
def test_native_group_equals_single_domain_bitwise(canvas, cfg, halo, world, steps, schedule, overlap):
    """fluid_group_step_n: the same plan, windowed kernels and ghost-row addressing as the RCCL driver, the whole stripe
    set in this process with device-to-device copies for the exchanges — bitwise equal to the single-domain run"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=schedule, random=fluid_hip.mulberry32(9)) as one:
        one.multipleSplats(6)
        one.step(0.016666, steps)
        want = one.fields()
    g = StripeGroup(world, canvas=canvas, config=cfg, halo=halo, schedule=schedule, random=fluid_hip.mulberry32(9), overlap=overlap)
    try:
        g.multipleSplats(6)
        g.step(0.016666, steps)
        g.sync()
        g.check_halo()
        for k in S.FIELDS:
            got = g.read(k)
            assert got.shape == want[k].shape
            assert np.array_equal(got, want[k]), k
        plan = fluid_hip._abi.stripe_plan(halo, g.engines[0].info("dye").halo, cfg["PRESSURE_ITERATIONS"])
        assert g.exchanges == steps * sum(1 for op in plan if op[0] == "exchange")
    finally:
        g.close()


def test_native_group_halo_overflow_raises():
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    g = StripeGroup(2, canvas=(256, 256), config={"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 4}, halo=4)
    try:
        g.splat(0.5, 0.5, 0.0, 90000.0, (1, 1, 1))
        with pytest.raises(fluid_hip.FluidError) as e:
            g.step(0.016666)     # the step that samples a row / column that was not refreshed fails itself (FLUID_ERR_HALO) ...
        assert e.value.status == -5
        g.check_halo()           # ... and resets the counter
    finally:
        g.close()


@pytest.mark.parametrize("overlap", [True, False])
def test_native_group_back_trace_beyond_reach_is_reported(overlap):
    """only `reach` ghost rows are refreshed before the advection: a longer back-trace must not be served from a stale
    (or in-flight) row — it is counted, and the step raises"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 10}
    g = StripeGroup(2, canvas=(256, 256), config=cfg, halo=32, reach=3, overlap=overlap)
    try:
        assert g.engines[0].advect_exchange_rows() == (3, 3)
        g.splat(0.5, 0.5, 0.0, 800.0, (1, 1, 1))      # dt * |v| = 13 rows > reach 3, < halo 32
        with pytest.raises(fluid_hip.FluidError) as e:
            g.step(0.016666)     # the step that samples a row / column that was not refreshed fails itself (FLUID_ERR_HALO) ...
        assert e.value.status == -5
        g.check_halo()           # ... and resets the counter
    finally:
        g.close()


def test_native_group_default_reach_covers_the_velocity_clamp():
    """|v| <= 1000 at the vorticity clamp (script.js:864), a projection overshoot on top, dt <= 1/60 (script.js:1191): the
    default reach of 24 rows covers back-traces of up to 22 rows plus the bilinear footprint"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 10, "CURL": 0}
    with fluid_hip.FluidSim(canvas=(256, 256), config=cfg) as one:
        one.splat(0.5, 0.5, 0.0, 5000.0, (1, 1, 1)); one.splat(0.5, 0.5, 3000.0, -5000.0, (1, 0, 1))
        one.step(0.016666, 2)
        want = one.fields()
    g = StripeGroup(2, canvas=(256, 256), config=cfg, halo=32)
    try:
        g.splat(0.5, 0.5, 0.0, 5000.0, (1, 1, 1)); g.splat(0.5, 0.5, 3000.0, -5000.0, (1, 0, 1))
        g.step(0.016666, 2)
        g.check_halo()
        for k in S.FIELDS:
            assert np.array_equal(g.read(k), want[k]), k
    finally:
        g.close()


def test_stripe_without_communicator_fails_loudly():
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    g = StripeGroup(2, canvas=(256, 256), config={"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64}, halo=8)
    try:
        with pytest.raises(fluid_hip.FluidError) as e:
            g.engines[0].step_n(1, 0.016666, g.config)     # a lone stripe cannot step: no RCCL communicator
        assert e.value.status == -7
    finally:
        g.close()


def test_rccl_communicator_and_stream_ordered_self_exchange():
    """RCCL resolved at run time, ncclCommInitRank, and a grouped ncclSend/ncclRecv pair ordered on the context
    stream between two kernels.  One rank is all a single-GPU box can host; the N-rank run is bench.py --gpus N."""
    import fluid_hip
    from fluid_hip.stripes import HipStripeEngine, new_comm_id
    from fluid_hip import _abi
    e = HipStripeEngine((64, 64), (64, 64), 0, 1, 0, _abi.SCHED_FUSED, 0)
    try:
        e.use_own_stream()
        e.comm_init(new_comm_id())
        e.comm_selftest(1 << 18)
        e.step_n(2, 0.016666, fluid_hip.DEFAULT_CONFIG)   # parts == 1: the whole-domain step, unaffected by the communicator
        e.sync()
    finally:
        e.close()


def fake_rccl_lib():
    """tests/fake_rccl/libfake_rccl.so, rebuilt when its source is newer (the in-process stand-in for librccl)"""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    lib, src = os.path.join(here, "fake_rccl", "libfake_rccl.so"), os.path.join(here, "fake_rccl", "fake_rccl.cpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["bash", os.path.join(here, "fake_rccl", "build.sh")], stdout=subprocess.DEVNULL)
    return lib


# ---- the RCCL leg with SEVERAL ranks: rank threads in a fresh process, the in-process stand-in loaded as librccl ----------
@pytest.mark.parametrize("world,halo,overlap,cfg,canvas,steps", [
    (2, 56, True, {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}, (512, 512), 2),
    (4, 32, True, {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}, (512, 512), 2),
    (4, 32, False, {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 50}, (512, 512), 2),
    (8, 12, True, {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 30}, (256, 1024), 1),
    (3, 16, True, {"SIM_RESOLUTION": 96, "DYE_RESOLUTION": 192, "PRESSURE_ITERATIONS": 20}, (512, 512), 2),     # dye grid != sim grid
])
def test_native_rccl_path_with_several_ranks_bitwise(world, halo, overlap, cfg, canvas, steps):
    """fluid_step_n on stripe contexts with a communicator — stripe_step_n, rccl_exchange_begin / end, the interior-first
    overlap — run by `world` rank threads against tests/fake_rccl (an in-process implementation of the nccl point-to-point
    calls with NCCL's matching and ordering rules).  Assembled result bitwise equal to the single domain."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = fake_rccl_lib()
    args = {"world": world, "halo": halo, "overlap": overlap, "config": cfg, "canvas": list(canvas), "steps": steps}
    env = dict(os.environ, FLUID_RCCL_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(here, "fake_rccl", "run_ranks.py"), json.dumps(args)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out
    assert out["rccl"] == lib
    import fluid_hip
    dye_halo = -(-halo * cfg["DYE_RESOLUTION"] // cfg["SIM_RESOLUTION"])
    va = min(24 + (0 if cfg["DYE_RESOLUTION"] == cfg["SIM_RESOLUTION"] else 1), halo)       # default reach: 24 rows
    plan = fluid_hip._abi.stripe_plan(halo, dye_halo, cfg["PRESSURE_ITERATIONS"], va, min(dye_halo, va if cfg["DYE_RESOLUTION"] == cfg["SIM_RESOLUTION"] else 49))
    assert out["exchanges"] == steps * sum(1 for op in plan if op[0] == "exchange")


@pytest.mark.parametrize("world,tiles_x,canvas,res", [(3, 1, (256, 768), 256), (4, 2, (512, 512), 512)])
def test_link_calibration_between_rank_threads_then_steps_bitwise(world, tiles_x, canvas, res):
    """fluid_comm_calibrate_link as `bench.py --gpus N` runs it: every rank thread of a set (tests/fake_rccl, real matching between the
    ranks, not loopback) probes its neighbours right behind comm_init — 2 x 23 grouped exchanges with up to four of them — and the set then
    steps: the probe must leave every pair's send / receive order in step (else the first exchange hangs or mismatches) and hand every rank
    a model; the fields are still the single domain's, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = fake_rccl_lib()
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": res, "PRESSURE_ITERATIONS": 30}
    args = {"world": world, "tiles_x": tiles_x, "halo": 24, "overlap": True, "config": cfg, "canvas": list(canvas), "steps": 2, "calibrate": True}
    r = subprocess.run([sys.executable, os.path.join(here, "fake_rccl", "run_ranks.py"), json.dumps(args)], env=dict(os.environ, FLUID_RCCL_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out
    assert len(out["links"]) == world and all(l is not None and l[0] >= 0 and l[1] > 0 for l in out["links"]), out["links"]


# ---- 2-D tile decomposition (BASELINE configs[3]: 2 x 2 on four GPUs): ghost columns as well -------------------------------
TILE_CASES = [
    # canvas, config, halo, tiles_y, tiles_x, steps
    ((512, 512), {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}, 56, 2, 2, 2),
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 50}, 32, 2, 2, 2),
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 20}, 16, 1, 4, 2),   # column tiles only
    ((1024, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 30}, 24, 2, 4, 1),  # 512 x 256 grid, 2 x 4 tiles
    ((512, 512), {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 20}, 16, 2, 2, 2),   # dye grid != sim grid
    ((512, 512), {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 0, "CURL": 0}, 8, 2, 2, 2),
]


@pytest.mark.parametrize("schedule,overlap", [("fused", True), ("fused", False), ("passes", True)])
@pytest.mark.parametrize("canvas,cfg,halo,ty,tx,steps", TILE_CASES)
def test_native_tile_group_equals_single_domain_bitwise(canvas, cfg, halo, ty, tx, steps, schedule, overlap):
    """tiles_y x tiles_x contexts in one process (fluid_group_step_n): ghost columns between left / right neighbours, then
    ghost rows including the fresh ghost columns (corners without diagonal messages) — bitwise equal to the single domain"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=schedule, random=fluid_hip.mulberry32(9)) as one:
        one.multipleSplats(6)
        one.step(0.016666, steps)
        want = one.fields()
    g = StripeGroup(ty * tx, canvas=canvas, config=cfg, halo=halo, schedule=schedule, random=fluid_hip.mulberry32(9), tiles_x=tx,
                    overlap=overlap)
    try:
        g.multipleSplats(6)
        g.step(0.016666, steps)
        g.sync()
        g.check_halo()
        for k in S.FIELDS:
            got = g.read(k)
            assert got.shape == want[k].shape
            assert np.array_equal(got, want[k]), k
    finally:
        g.close()


def test_native_tile_group_back_trace_beyond_reach_is_reported():
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 10}
    g = StripeGroup(2, canvas=(256, 256), config=cfg, halo=32, reach=3, tiles_x=2)     # 1 x 2: only ghost columns
    try:
        g.splat(0.5, 0.5, 800.0, 0.0, (1, 1, 1))       # a horizontal jet across the tile border: dt * |v| = 13 columns > 3
        with pytest.raises(fluid_hip.FluidError) as e:
            g.step(0.016666)     # the step that samples a row / column that was not refreshed fails itself (FLUID_ERR_HALO) ...
        assert e.value.status == -5
        g.check_halo()           # ... and resets the counter
    finally:
        g.close()


@pytest.mark.parametrize("ty,tx,halo,cfg", [(2, 2, 56, {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}),
                                            (2, 2, 16, {"SIM_RESOLUTION": 128, "DYE_RESOLUTION": 256, "PRESSURE_ITERATIONS": 20}),
                                            (1, 3, 16, {"SIM_RESOLUTION": 192, "DYE_RESOLUTION": 192, "PRESSURE_ITERATIONS": 20}),
                                            # a centre tile with all EIGHT neighbours; halo 24: three pressure blocks, the exchanges between them
                                            # and the step's first one covered by cut Jacobi launches (two of them at this size)
                                            (3, 3, 24, {"SIM_RESOLUTION": 384, "DYE_RESOLUTION": 384, "PRESSURE_ITERATIONS": 50})])
def test_native_rccl_path_2d_tiles_with_several_ranks_bitwise(ty, tx, halo, cfg):
    """the RCCL leg of the 2-D decomposition (ONE grouped round to up to eight neighbours — sides and corners —, staged blocks, one
    message per neighbour) with ty x tx rank threads against tests/fake_rccl"""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = fake_rccl_lib()
    args = {"world": ty * tx, "tiles_x": tx, "halo": halo, "config": cfg, "canvas": [512, 512], "steps": 2}
    r = subprocess.run([sys.executable, os.path.join(here, "fake_rccl", "run_ranks.py"), json.dumps(args)], env=dict(os.environ, FLUID_RCCL_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out


def _random_decompositions(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        ty, tx = int(rng.choice([1, 2, 3, 4, 8])), int(rng.choice([1, 1, 2, 3, 4]))
        if ty * tx < 2 or ty * tx > 12:
            continue
        halo = int(rng.integers(3, 11)) * 4                              # 12 … 40
        rows, cols = int(rng.integers(halo, 4 * halo)), int(rng.integers(max(halo // 4, 6), 60)) * 4
        w, h = cols * tx, rows * ty
        if (cols < halo and tx > 1) or w < 16 or h < 16 or w > 1400 or h > 1400:
            continue
        res = min(w, h)
        same = rng.random() < 0.7
        dye = res if same else res * 2
        out.append(((w, h), {"SIM_RESOLUTION": res, "DYE_RESOLUTION": dye, "PRESSURE_ITERATIONS": int(rng.integers(0, 61)),
                             "CURL": float(rng.choice([0.0, 30.0]))}, halo, ty, tx, int(rng.integers(1, 3))))
    return out


@pytest.mark.parametrize("canvas,cfg,halo,ty,tx,steps", _random_decompositions(24, 7))
def test_native_random_decompositions_equal_single_domain_bitwise(canvas, cfg, halo, ty, tx, steps):
    """randomised stripe / tile sets: ragged tile sizes, halo 12 … 40, 0 … 60 Jacobi iterations (1 … 7 pressure blocks),
    dye grid = or 2 x the sim grid, 1-D and 2-D — the decomposed result equals the single domain bit for bit"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, random=fluid_hip.mulberry32(3)) as one:
        one.multipleSplats(5)
        one.step(0.016666, steps)
        want = one.fields()
    g = StripeGroup(ty * tx, canvas=canvas, config=cfg, halo=halo, random=fluid_hip.mulberry32(3), tiles_x=tx)
    try:
        g.multipleSplats(5)
        g.step(0.016666, steps)
        g.check_halo()
        for k in S.FIELDS:
            assert np.array_equal(g.read(k), want[k]), (k, canvas, cfg, halo, ty, tx)
    finally:
        g.close()


@pytest.mark.gpu
def test_overlap_probe_runs_one_rank_in_loopback():
    """tools/overlap_vs_link.py (the probe behind profiles/r04/overlap_vs_link_latency.txt): one rank of three in tests/fake_rccl's loopback
    mode, with a link that takes time — the child must step and report, and a 200 us link with the overlap off must cost at least the two
    exchanges' 400 us per step (the probe's own check that the delay is real)"""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    tool = os.path.join(os.path.dirname(here), "tools", "overlap_vs_link.py")
    lib = fake_rccl_lib()
    out = {}
    for delay in (0, 200):
        env = dict(os.environ, FLUID_RCCL_LIB=lib, FAKE_RCCL_LOOPBACK="1", _OVL_CHILD=json.dumps({"config": "stripe", "overlap": 0}))
        env.pop("FAKE_RCCL_GBPS", None)
        env["FAKE_RCCL_DELAY_US"] = str(delay)
        r = subprocess.run([sys.executable, tool], env=env, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert lines, r.stderr[-500:]
        out[delay] = json.loads(lines[-1])
        assert out[delay]["ok"] and out[delay]["exchanges_per_step"] == 2.0
    assert out[200]["ms_per_step"] - out[0]["ms_per_step"] > 0.35, out


@pytest.mark.parametrize("link", [(0.0, 1e6), (5000.0, 1.0)])
@pytest.mark.parametrize("ty,tx,halo,iters", [(4, 1, 24, 50), (2, 2, 56, 50), (3, 3, 16, 30)])
def test_link_model_changes_the_schedule_not_the_bits(ty, tx, halo, iters, link):
    """fluid_set_link_model sizes how many leading Jacobi launches are cut around an exchange (an instantaneous link: none behind the
    step's first exchange, one behind a pressure exchange; a very slow one: two wherever the tile is large enough) — speed only: the
    decomposed run equals the single domain bit for bit either way"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 576, "DYE_RESOLUTION": 576, "PRESSURE_ITERATIONS": iters}
    with fluid_hip.FluidSim(canvas=(576, 576), config=cfg, random=fluid_hip.mulberry32(11)) as one:
        one.multipleSplats(6)
        one.step(0.016666, 2)
        want = one.fields()
    g = StripeGroup(ty * tx, canvas=(576, 576), config=cfg, halo=halo, random=fluid_hip.mulberry32(11), tiles_x=tx, link_model=link)
    try:
        g.multipleSplats(6)
        g.step(0.016666, 2)
        g.check_halo()
        for k in S.FIELDS:
            assert np.array_equal(g.read(k), want[k]), (k, ty, tx, link)
    finally:
        g.close()


@pytest.mark.parametrize("delay_us,gbps", [(60, 25), (80, 50), (150, 100)])
def test_link_calibration_finds_the_link_it_is_given(delay_us, gbps):
    """fluid_comm_calibrate_link against tests/fake_rccl with a synthetic link (latency + bytes / bandwidth per exchange): the measured model
    is within 20 % of the injected one.  One rank of three in the stand-in's loopback mode (the middle stripe: two neighbours), its own
    process per link because the stand-in reads its environment once.  (The stand-in itself costs 7 us per exchange — a spin kernel and two
    device-to-device copies, which the probe rightly counts: profiles/r05/link_calibration_repeat.txt — so the slowest link here is 60 us.)  (The defaults the probe replaces — 20 us, 50 GB/s — decide how many
    Jacobi launches the driver cuts around an exchange; a wrong guess costs 2-3 % either way: profiles/r04/overlap_vs_link_latency.txt.)"""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    tool = os.path.join(os.path.dirname(here), "tools", "overlap_vs_link.py")
    lib = fake_rccl_lib()
    env = dict(os.environ, FLUID_RCCL_LIB=lib, FAKE_RCCL_LOOPBACK="1", FAKE_RCCL_DELAY_US=str(delay_us), FAKE_RCCL_GBPS=str(gbps),
               _OVL_CHILD=json.dumps({"config": "stripe", "overlap": 1, "calibrate": True}))
    r = subprocess.run([sys.executable, tool], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stderr[-800:]
    d = json.loads(lines[-1])
    assert abs(d["latency_us"] - delay_us) <= 0.2 * delay_us, d
    assert abs(d["GBps"] - gbps) <= 0.2 * gbps, d


@pytest.mark.parametrize("world,tiles_x,canvas", [(2, 1, (4096, 4608)), (4, 2, (6144, 6144))])
def test_packed_dye_on_a_stripe_or_tile_set_leaves_the_same_bits(world, tiles_x, canvas):
    """Round 5: stripe / tile contexts of at least 3072^2 owned texels keep their dye PACKED (three floats per texel) through the fused
    advection like a whole domain does, and the ghost texels travel as 12-byte texels, in place.  The set must leave what the single domain
    leaves, bit for bit — through splats into a packed field, a READ of one context only in mid-run (which must not change that context's
    format: the format is part of the message layout both ends of an exchange cut), and steps behind it."""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    res = min(canvas)
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": res, "PRESSURE_ITERATIONS": 12}
    DT = 0.016666
    g = StripeGroup(world, canvas=canvas, config=cfg, halo=24, random=fluid_hip.mulberry32(77), tiles_x=tiles_x)
    try:
        with fluid_hip.FluidSim(canvas=canvas, config=cfg, random=fluid_hip.mulberry32(77)) as one:
            for sim in (g, one):
                sim.multipleSplats(4)
                sim.step(DT, 3)
            info = [e.schedule_info(3, DT, g.config) for e in g.engines]
            assert all(i["dye_packed"] for i in info), info               # the set really runs the packed path
            peek = g.engines[0].read("dye")                               # ONE context read: converts into the spare buffer, changes nothing
            assert peek[..., 3].min() == peek[..., 3].max() and 0.9 < float(peek[0, 0, 3]) < 1.0   # the decayed alpha, one value
            assert g.engines[0].schedule_info(3, DT, g.config)["dye_packed"]
            for sim in (g, one):
                sim.multipleSplats(2)                                     # splats into the packed field
                sim.step(DT, 4)
            for k in ("velocity", "pressure", "divergence", "curl", "dye"):
                assert np.array_equal(g.read(k), one.read(k)), k
    finally:
        g.close()


@pytest.mark.parametrize("world,tiles_x,canvas,rank", [(2, 1, (4096, 4608), 0), (3, 1, (4096, 6912), 1)])
def test_ranks_agree_on_the_dye_wire_format_when_one_of_them_loses_its_alpha(world, tiles_x, canvas, rank):
    """the dye's ghost texels travel as 12-byte texels while the field is packed and as RGBA otherwise, and whether a rank packs
    hangs on state ONE rank can change alone between two calls (here: a raw device pointer to its dye, which takes the known alpha away).  Over
    RCCL two neighbours would then post ncclSend / ncclRecv of different sizes — undefined (the stand-in refuses the count mismatch: this test
    failed with "invalid argument" before round 6).  Every fluid_step_n on a communicator now opens with one all-reduce of four floats in which
    the set agrees on the call's format: the ranks step packed, one of them is touched alone, the next call runs RGBA on ALL of them — and the
    set leaves the single domain's bits."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = fake_rccl_lib()
    res = min(canvas)
    args = {"world": world, "tiles_x": tiles_x, "halo": 24, "overlap": True, "canvas": list(canvas), "steps": 2,
            "config": {"SIM_RESOLUTION": res, "DYE_RESOLUTION": res, "PRESSURE_ITERATIONS": 12}, "lone_touch": {"rank": rank}}
    r = subprocess.run([sys.executable, os.path.join(here, "fake_rccl", "run_ranks.py"), json.dumps(args)], env=dict(os.environ, FLUID_RCCL_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"], out
    # every rank reported "packed" behind the first call and "not packed" behind the second (2 * world answers, in whatever order the threads ran)
    assert sorted(out["packed"]) == [False] * world + [True] * world, out


def test_a_set_that_disagrees_on_the_dye_is_refused():
    """what "splats are collective on a set" means where one process can see it: a splat into ONE context of an in-process set leaves the
    contexts with different alphas (1 against the decayed value) — fluid_group_step_n refuses to exchange instead of mixing texel formats"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    cfg = {"SIM_RESOLUTION": 4096, "DYE_RESOLUTION": 4096, "PRESSURE_ITERATIONS": 4}
    g = StripeGroup(2, canvas=(4096, 4608), config=cfg, halo=24, random=fluid_hip.mulberry32(5))
    try:
        g.multipleSplats(2)
        g.step(0.016666, 2)
        g.engines[1].splat(0.5, 0.5, 10.0, 10.0, 1.0, 0.5, 0.25, 4096 / 4608, 0.0025)
        with pytest.raises(fluid_hip.FluidError) as e:
            g.step(0.016666, 1)
        assert "collective" in str(e.value)
    finally:
        g.close()


def test_link_model_rejects_nonsense():
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    g = StripeGroup(2, canvas=(256, 256), config={"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 256}, halo=16)
    try:
        with pytest.raises(fluid_hip.FluidError):
            g.engines[0].set_link_model(-1.0, 50.0)
        with pytest.raises(fluid_hip.FluidError):
            g.engines[0].set_link_model(20.0, 0.0)
    finally:
        g.close()
