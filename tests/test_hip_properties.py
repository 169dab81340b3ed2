"""Size-independent properties at BASELINE.json's full sizes (where the CPU oracle is too slow to be
the checker) plus schedule equivalence: the fused schedule must reproduce the per-pass schedule bit
for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sim_of(N, schedule, iters=50, **cfg):
    import fluid_hip
    c = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": iters}
    c.update(cfg)
    return fluid_hip.FluidSim(canvas=(N, N), config=c, schedule=schedule, random=fluid_hip.mulberry32(1234))


@pytest.mark.parametrize("N", [1024, 4096])
def test_fused_equals_passes_bitwise(N):
    a, b = sim_of(N, "passes"), sim_of(N, "fused")
    try:
        a.multipleSplats(8); b.multipleSplats(8)
        a.step(0.016666, 2); b.step(0.016666, 2)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(a.read(k), b.read(k)), k
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("canvas,res,dye", [((512, 512), 64, 64), ((520, 300), 300, 300), ((1000, 40), 40, 40), ((4096, 128), 128, 128),
                                            ((256, 3000), 256, 256), ((640, 480), 240, 480), ((250, 130), 130, 130),
                                            # widths that are not multiples of 4 (most of what getResolution, script.js:1612-1624, produces):
                                            # 1001 x 300, 303 x 128 (the 2560 x 1080 canvas at SIM_RESOLUTION 128), 257 x 64 (one tile + 1 column),
                                            # 1366 x 768 -> 455 x 256 with a 910 x 512 dye grid, and grids narrower than one quad
                                            ((1001, 300), 300, 300), ((2560, 1080), 128, 128), ((257, 64), 64, 64), ((1366, 768), 256, 512),
                                            ((3, 64), 3, 3), ((64, 5), 5, 5), ((1, 1), 1, 1)])
def test_fused_equals_passes_bitwise_odd_shapes(canvas, res, dye):
    """tile-boundary coverage of the fused kernels: widths/heights that are not multiples of the tile, grids
    smaller than one tile, W % 4 != 0 (the last quad of a row is partly padding), dye grid != sim grid — and the fused
    schedule really runs its register-tile kernels at every width (launch counts)"""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": dye, "PRESSURE_ITERATIONS": 23}
    sims = [fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=s, random=fluid_hip.mulberry32(5)) for s in ("passes", "fused")]
    try:
        for s in sims:
            s.multipleSplats(7)
            s.step(0.016666, 3)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
        for s, launches in zip(sims, (23, 3)):     # 23 iterations: one launch each, or three temporally blocked launches of <= 10
            s.set_timing(True)
            s.step(0.016666, 2)
            s.sync()
            t = s.timings()
            s.set_timing(False)
            assert t["steps"] == 2 and t["jacobi_launches"] == 2 * launches, (s.schedule if hasattr(s, "schedule") else "", t)
            if launches == 3:
                assert t["curl_ms"] == 0 and t["divergence_ms"] == 0     # curl + vorticity + divergence ran as ONE kernel
    finally:
        for s in sims:
            s.close()


@pytest.mark.parametrize("vd,dd,dt", [(0.2, 1.0, 0.016666),      # the defaults: decays 1.003 / 1.017 -> k_advect_both_fast (div_uniform)
                                      (0.0, 0.0, 0.016666),      # decay exactly 1
                                      (4.0, 4.0, 0.016666),      # the GUI's maxima
                                      (59.0, 0.3, 0.016666),     # velocity decay 1.98: still the fast kernel; heavy damping -> subnormal quotients
                                      (100.0, 1.0, 0.016666),    # decay 2.67: outside [1, 2) -> the general kernel
                                      (-0.5, -2.0, 0.016666),    # decays below 1 (growth): the general kernel
                                      (0.2, 1.0, 0.0)])          # dt = 0
def test_fused_advection_divides_exactly_whatever_the_decay(vd, dd, dt):
    """the fused advection kernel replaces the IEEE divide by a wave-uniform divisor with a double multiply (fluid_math.h div_uniform)
    where that is provably exact, and falls back to the general kernel elsewhere: either way the bits of the per-pass kernels, which
    divide the plain way — incl. tiny values that decay into the subnormal range (dye tails of exp_reference are ~1e-38 from the start)"""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": 250, "DYE_RESOLUTION": 250, "PRESSURE_ITERATIONS": 12, "VELOCITY_DISSIPATION": vd, "DENSITY_DISSIPATION": dd}
    sims = [fluid_hip.FluidSim(canvas=(500, 250), config=cfg, schedule=s, random=fluid_hip.mulberry32(11)) for s in ("passes", "fused")]
    try:
        rng = np.random.default_rng(3)
        H, W = sims[0].velocity.height, sims[0].velocity.width
        tiny = (rng.normal(0, 1, (H, W, 4)) * rng.choice([1e-44, 1e-40, 1e-38, 1e-30, 1.0], (H, W, 4))).astype(np.float32)
        for s in sims:
            s.multipleSplats(5)
            s.write("dye", np.abs(tiny) + s.read("dye") * (np.abs(tiny) > 0.5))   # mostly subnormal / tiny dye, some ordinary values
            s.step(dt, 4)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
        d = sims[1].read("dye")
        assert np.isfinite(d).all() and ((d != 0) & (np.abs(d) < 1.2e-38)).any()          # subnormals really went through the divide
    finally:
        for s in sims:
            s.close()


def test_zero_state_is_a_fixed_point_4096():
    s = sim_of(4096, "fused")
    try:
        s.step(0.016666, 3)
        assert not s.read("velocity").any() and not s.read("pressure").any()
        d = s.read("dye")
        assert not d[..., :3].any()
        # fresh alpha 1 decays by 1/(1 + DENSITY_DISSIPATION*dt) per step (script.js:779-780)
        a = np.float32(1.0)
        for _ in range(3):
            a = a / (np.float32(1.0) + np.float32(1.0) * np.float32(0.016666))
        assert np.all(d[..., 3] == a)
    finally:
        s.close()


def test_centred_splat_mirror_symmetry_4096():
    s = sim_of(4096, "fused", CURL=0)
    try:
        s.splat(0.5, 0.5, 0.0, 700.0, {"r": 1.0, "g": 0.5, "b": 0.25})
        s.step(0.016666, 2)
        v, d, p = s.read("velocity"), s.read("dye"), s.read("pressure")
    finally:
        s.close()
    # an upward jet through the centre is mirror-symmetric in x: vy, dye, p even; vx odd
    assert np.allclose(v[..., 1], v[:, ::-1, 1], rtol=0, atol=2e-4 * np.abs(v).max())
    assert np.allclose(v[..., 0], -v[:, ::-1, 0], rtol=0, atol=2e-4 * np.abs(v).max())
    assert np.allclose(d, d[:, ::-1], rtol=0, atol=2e-4 * np.abs(d).max())
    assert np.allclose(p, p[:, ::-1], rtol=0, atol=2e-4 * np.abs(p).max())


def test_jacobi_residual_decreases_4096():
    s = sim_of(4096, "fused", iters=0)
    try:
        s.multipleSplats(10)
        s.run_pass("divergence")
        div = s.read("divergence").astype(np.float64)

        def residual():
            p = np.pad(s.read("pressure").astype(np.float64), 1, mode="edge")
            lap = p[1:-1, :-2] + p[1:-1, 2:] + p[:-2, 1:-1] + p[2:, 1:-1] - 4 * p[1:-1, 1:-1]
            return float(np.abs(lap - div).mean())

        r0 = residual()
        s.run_pass("jacobi", iters=8)
        r1 = residual()
        s.run_pass("jacobi", iters=42)
        r2 = residual()
    finally:
        s.close()
    assert r2 < r1 < r0


def test_step_n_equals_n_steps():
    a, b = sim_of(512, "fused", iters=20), sim_of(512, "fused", iters=20)
    try:
        a.multipleSplats(5); b.multipleSplats(5)
        a.step(0.016666, 4)
        for _ in range(4):
            b.step(0.016666)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(a.read(k), b.read(k)), k
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("canvas,res,n", [((5, 3), 3, 3), ((63, 64), 64, 2), ((64, 58), 58, 3), ((65, 59), 59, 4), ((130, 7), 7, 5), ((57, 300), 57, 3),
                                          ((1001, 700), 700, 3), ((2048, 2048), 2048, 2), ((4096, 4096), 4096, 3)])
def test_a_call_for_n_steps_leaves_what_n_calls_leave(canvas, res, n):
    """fluid_step_n(n > 1) hands each step's advected velocity to the next step's curl / vorticity / divergence inside one launch
    (k_advect_cvd: 64-column tiles of one texel per lane); every field afterwards — the curl and divergence of the LAST step included —
    must be what n separate calls leave, and what the per-pass schedule leaves.  Shapes: narrower / lower than a tile, one texel past a tile,
    widths that are no multiple of 4, the bench grid."""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": res, "PRESSURE_ITERATIONS": 12}
    sims = [fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=s, random=fluid_hip.mulberry32(31)) for s in ("fused", "fused", "passes")]
    try:
        for s in sims:
            s.multipleSplats(6)
        sims[0].step(0.016666, n)
        sims[2].step(0.016666, n)
        for _ in range(n):
            sims[1].step(0.016666)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            ref = sims[1].read(k)
            assert np.array_equal(sims[0].read(k), ref), k
            assert np.array_equal(sims[2].read(k), ref), k
        sims[0].step(0.016666, 2)        # and the chain starts from whatever the last call left
        sims[1].step(0.016666); sims[1].step(0.016666)
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
    finally:
        for s in sims:
            s.close()


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        w = int(rng.integers(1, 180)) * 4 if rng.random() < 0.8 else int(rng.integers(5, 700))     # mostly W % 4 == 0 (fused kernels)
        h = int(rng.integers(4, 700))
        res = min(w, h)
        canvas = (w, h)
        dye = res if rng.random() < 0.6 else int(res * rng.choice([0.5, 1.5, 2.0]))
        cases.append((canvas, res, max(dye, 4), int(rng.integers(0, 61)), float(rng.choice([0.0, 30.0, 55.5])), int(rng.integers(1, 4))))
    return cases


@pytest.mark.parametrize("canvas,res,dye,iters,curl,steps", _random_cases(40, 2024))
def test_fused_equals_passes_bitwise_random_shapes(canvas, res, dye, iters, curl, steps):
    """randomised differential test of the two schedules: ragged widths / heights around the tile sizes (256 x 40 and
    256 x 80 texel tiles, aprons 3 / 4 and 10 / 12), grids smaller than a tile, 0 … 60 Jacobi iterations, dye != sim"""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": dye, "PRESSURE_ITERATIONS": iters, "CURL": curl}
    sims = [fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule=s, random=fluid_hip.mulberry32(17)) for s in ("passes", "fused")]
    try:
        for s in sims:
            s.multipleSplats(5)
            s.step(0.016666, steps)
        assert [sims[0].velocity.width, sims[0].velocity.height] == [sims[1].velocity.width, sims[1].velocity.height]
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), (k, canvas, cfg)
    finally:
        for s in sims:
            s.close()


# ---- every A/B knob of the fused schedule produces the same bits (each knob is read once per process: one child per setting) ----
_KNOB_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import fluid_hip
out = {}
for name, canvas, cfg in (("wide", (1001, 700), {"SIM_RESOLUTION": 700, "DYE_RESOLUTION": 700, "PRESSURE_ITERATIONS": 50}),
                          ("dye_ne_sim", (1024, 1024), {"SIM_RESOLUTION": 200, "DYE_RESOLUTION": 1024, "PRESSURE_ITERATIONS": 23})):
    with fluid_hip.FluidSim(canvas=canvas, config=cfg, schedule="fused", random=fluid_hip.mulberry32(77)) as sim:
        sim.multipleSplats(8)
        sim.step(0.016666, 3)
        h = hashlib.sha256()
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            h.update(sim.read(k).tobytes())
        out[name] = h.hexdigest()
print(json.dumps(out))
"""


@pytest.mark.parametrize("N,iters", [(1024, 20), (4096, 10), (4096, 3)])
def test_jacobi_tiny_values_fused_equals_passes(N, iters):
    """The temporally blocked Jacobi kernel against the per-pass one on fields whose left part is ordinary and whose right part lives around
    1e-36 .. 1e-45: results that are subnormal AND inexact are where (s - div) * 0.25 and a fused multiply-add part ways, so this is the
    case any rewrite of the update has to get through (profiles/r03/jacobi_fma_probe.txt: the fused form is 2 % faster and was not
    shipped, because nothing on gfx950 tells a tile for free that it is safe — TRAPSTS.EXCP stays 0, tools/micro/trapsts_probe.hip)."""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": iters, "CURL": 0}
    sims = [fluid_hip.FluidSim(canvas=(N, N), config=cfg, schedule=s, random=fluid_hip.mulberry32(5)) for s in ("passes", "fused")]
    try:
        rng = np.random.default_rng(N + iters)
        H, W = sims[0].velocity.height, sims[0].velocity.width
        scale = np.ones((H, W, 1), np.float32)
        scale[:, W // 3:] = rng.choice(np.array([1e-36, 1e-37, 3e-38, 1e-38, 1e-39, 1e-41, 1e-44], np.float32), (H, W - W // 3, 1))
        scale[H // 2:H // 2 + 37, :W // 3] = 1e-38                            # and a thin tiny band inside the ordinary part
        vel = (rng.normal(0, 1, (H, W, 2)).astype(np.float32) * scale).astype(np.float32)
        prs = (rng.normal(0, 1, (H, W)).astype(np.float32) * scale[..., 0]).astype(np.float32)
        for s in sims:
            s.write("velocity", vel)
            s.write("pressure", prs)
            s.step(0.016666, 1)
        a, b = sims[0].read("pressure"), sims[1].read("pressure")
        assert np.array_equal(a, b)
        for k in ("velocity", "divergence"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
        tiny = (b != 0) & (np.abs(b) < 1.1754944e-38)
        assert tiny.any() and (np.abs(b) > 1e-3).any()                         # subnormal pressures and ordinary ones came out of the same launch
    finally:
        for s in sims:
            s.close()


@pytest.mark.gpu
def test_every_knob_of_the_fused_schedule_yields_the_same_bits():
    """Jacobi tile shapes (incl. the deep small-grid ones and their gradient-subtract instantiations), K6 folded or not, the fast or
    the general advection kernels (dye grid == and != sim grid): all settings must hash to the same fields — at a width that is not a
    multiple of 4 and with the dye grid five times the sim grid."""
    import json
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "webgl-fluid-simulation_amd")
    settings = [{}] + [{"FLUID_TB_VARIANT": str(v), "FLUID_FOLD_GRADSUB": f} for v in (0, 8, 9, 10, 11, 12, 13, 14, 15, 20) for f in ("0", "1")]
    settings += [{"FLUID_TB_VARIANT": "0", "FLUID_FOLD_GRADSUB": "0", "FLUID_TB_TAIL": t} for t in ("0,130,5", "60,100,6", "100,0,7", "0,0,7", "60,100,2", "0,120,2")]   # small tiles for a launch's first / last rows (2 = the two-texel tile)
    settings += [{"FLUID_SKIP_CURL": "0"}]   # every step of a call for n steps stores its curl field, or only the last one
    settings += [{"FLUID_CVD_TAIL": t} for t in ("0,100", "60,90", "120,0")]   # the same for the curl / vorticity / divergence kernel
    settings += [{"FLUID_TB2": t, "FLUID_FOLD_GRADSUB": f} for t in ("8,4", "8,5", "8,6", "4,10", "16,3") for f in ("0", "1")]   # rows / waves of the two-texel Jacobi tile
    settings += [{"FLUID_CHAIN": "0"}] + [{"FLUID_CHAIN_TILE": t} for t in ("8,8,4", "4,8,4", "8,8,3", "16,8,4", "8,4,3", "4,4,3", "16,4,3")]   # advection + the next step's curl / vorticity / divergence in one launch, or not
    settings += [{"FLUID_ADVECT_WY": "2"}, {"FLUID_ADVECT_WY": "4"}, {"FLUID_ADVECT_WY": "4", "FLUID_ADVECT_ROWS": "2", "FLUID_ADVECT_SPLIT_ROWS": "4"}]   # the advection block's waves stacked in y
    # the dye != sim advection: velocity taps gathered (0) or read from the wave's LDS run (the product), one / two / four rows per thread
    settings += [{"FLUID_VTILE": "0"}, {"FLUID_VTILE": "0", "FLUID_ADVECT_SPLIT_ROWS": "4"}, {"FLUID_VTILE": "1", "FLUID_ADVECT_SPLIT_ROWS": "2"}]
    settings += [{"FLUID_ADVECT_FAST": "0"}, {"FLUID_ADVECT_SPLIT_ROWS": "4"}, {"FLUID_ADVECT_SPLIT_ROWS": "1"}, {"FLUID_ADVECT_ROWS": "2"},
                 {"FLUID_TB_VARIANT": "1", "FLUID_FOLD_GRADSUB": "1"}, {"FLUID_TB_VARIANT": "5"}]
    ref = None
    probes = os.path.join(pkg, "libfluid_hip_probes.so")   # make PROBES=1: the lab build, where the knobs are read at all
    assert os.path.exists(probes), "build the lab library first: make -C webgl-fluid-simulation_amd PROBES=1"
    for env in settings:
        if env:   # the first setting, {}, is the PRODUCT library: every lab shape has to reproduce ITS bits
            env = dict(env, FLUID_HIP_LIB=probes)
        r = subprocess.run([sys.executable, "-c", _KNOB_CHILD % pkg], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (env, r.stderr[-500:])
        got = json.loads(r.stdout.strip().splitlines()[-1])
        if ref is None:
            ref = got
        assert got == ref, (env, got, ref)


_CHAIN_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
import fluid_hip
cfg = {"SIM_RESOLUTION": 4096, "DYE_RESOLUTION": 4096, "PRESSURE_ITERATIONS": 20}
with fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, schedule="fused", random=fluid_hip.mulberry32(1234)) as sim:
    sim.multipleSplats(20)
    sim.step(0.016666, 4)
    print(json.dumps({k: hashlib.sha256(sim.read(k).tobytes()).hexdigest() for k in ("velocity", "pressure", "divergence", "curl", "dye")}))
"""


@pytest.mark.gpu
def test_the_chained_launch_at_the_bench_size_yields_the_same_bits():
    """above 3072^2 texels fluid_step_n keeps the separate advection and curl / vorticity / divergence launches by default
    (fluid_solver.cpp chain_enabled); FLUID_CHAIN=1 is the A/B knob bench visits use, so its bits are pinned at 4096^2 too"""
    import json
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "webgl-fluid-simulation_amd")
    got = []
    # ... and the other settings that only a big grid exercises: the two-texel tile as head / tail of the mixed Jacobi launch, the curl
    # field stored by every step of the call
    envs = [{"FLUID_CHAIN": "0"}, {"FLUID_CHAIN": "1"}, {"FLUID_TB_TAIL_TILES": "384,768,2"}, {"FLUID_SKIP_CURL": "0"}, {"FLUID_TB_TAIL_TILES": "0,0,7"},
            {"FLUID_DYE_PACK": "0"},   # the dye kept RGBA through the fused advection (the product packs it to three floats at this size)
            {"FLUID_JACOBI_CHAINS": "0.5"}, {"FLUID_JACOBI_CHAINS": "0.37"},   # the pressure loop as two row chains on two streams
            {"FLUID_ADVECT_XCD": "1"}, {"FLUID_ADVECT_WY": "4", "FLUID_ADVECT_XCD": "1"}, {"FLUID_ADVECT_WY": "2", "FLUID_ADVECT_ROWS": "2"},   # block shapes / XCD order of the packed-dye advection
            {"FLUID_CHAIN_GS": "1"}]   # K6 as one more block of the chained pressure launch (lab; profiles/r06/chain_gs_ab.txt: level to slower)
    probes = os.path.join(pkg, "libfluid_hip_probes.so")
    envs = [{}] + [dict(e, FLUID_HIP_LIB=probes) for e in envs]   # the product library's own bits first
    for env in envs:
        r = subprocess.run([sys.executable, "-c", _CHAIN_CHILD % pkg], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (env, r.stderr[-500:])
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    for env, g in zip(envs, got):
        assert g == got[0], env


@pytest.mark.gpu
def test_step_marks_and_schedule_info():
    """ABI 8: events between the steps of a call (nothing waits for them) and the description of what a call launches"""
    import fluid_hip
    DT = 0.016666
    cfg = {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 50}
    with fluid_hip.FluidSim(canvas=(512, 512), config=cfg, random=fluid_hip.mulberry32(5)) as sim:
        sim.multipleSplats(4)
        si = sim.schedule_info(6)
        assert si["fused"] == 1 and si["jacobi_launches"] == 5 and si["gradsub_folded"] == 1 and si["runs_ahead"] == 1 and si["pending_adopted"] == 0
        assert si["chained"] == 6   # every advection launch of the call also runs a curl / vorticity / divergence: five for the call's own steps, the last one ahead
        assert si["launches"] == 6 * 6 + 1   # five Jacobi launches (the last with K6) + advection (+ next curl) per step, one leading curl launch
        assert sim.schedule_info(1)["chained"] == 1 and sim.schedule_info(1)["launches"] == 7
        sim.set_step_marks(4)
        sim.step(DT, 6)                      # the first four steps are marked
        ms = sim.step_marks()
        assert len(ms) == 4 and all(0.0 < x < 50.0 for x in ms)
        sim.step(DT, 2)
        assert len(sim.step_marks()) == 2
        sim.set_step_marks(0)
        sim.step(DT, 2)
        assert sim.step_marks() == []
        want = sim.fields()
    with fluid_hip.FluidSim(canvas=(512, 512), config=cfg, random=fluid_hip.mulberry32(5)) as sim:   # marks change nothing
        sim.multipleSplats(4)
        sim.step(DT, 6); sim.step(DT, 2); sim.step(DT, 2)
        got = sim.fields()
    for k in want:
        assert np.array_equal(want[k], got[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("N,DYE", [(1024, 1024), (250, 250), (128, 1024), (303, 700)])
def test_one_step_per_call_works_ahead_and_leaves_what_the_passes_leave(N, DYE):
    """The per-frame path (update() -> step(dt) once per call, script.js:1176-1186): on grids where fluid_step_n chains, the launch that ends a
    call also runs the NEXT call's curl / vorticity / divergence into pending buffers; the next call adopts them unless something touched the
    fields or dt / CURL changed.  After EVERY call all five fields are what the per-pass schedule leaves — the advected velocity, this
    step's divergence and curl — with a splat, a dt change, a CURL change, a field write and a multi-step call in between."""
    import fluid_hip
    DT = 0.016666
    # (dye grid != sim grid — the reference's default shape — chains too since round 4: the velocity's advection + the next step's curl /
    # vorticity / divergence in one launch, the dye pass behind it; every step hands over through the pending buffers)
    side = max(N, DYE)
    mk = lambda sched: fluid_hip.FluidSim(canvas=(side, side), config={"SIM_RESOLUTION": N, "DYE_RESOLUTION": DYE, "PRESSURE_ITERATIONS": 50},
                                          schedule=sched, random=fluid_hip.mulberry32(1234))
    a, b = mk("passes"), mk("fused")
    adopted = []
    try:
        a.multipleSplats(5); b.multipleSplats(5)
        assert b.schedule_info(1)["runs_ahead"] == 1
        for k in range(12):
            dt, n = DT, 1
            if k == 3:
                for s_ in (a, b):
                    s_.splat(0.4, 0.6, 300.0, -200.0, {"r": 0.3, "g": 0.1, "b": 0.2})
            if k == 5:
                dt = 0.01
            if k == 7:
                a.config["CURL"] = b.config["CURL"] = 10
            if k == 9:
                v = b.read("velocity")
                a.write("velocity", v * 0.5); b.write("velocity", v * 0.5)
            if k == 10:
                n = 3
            adopted.append(b.schedule_info(n, dt)["pending_adopted"])
            a.step(dt, n); b.step(dt, n)
            for f in ("velocity", "pressure", "divergence", "curl", "dye"):
                assert np.array_equal(a.read(f), b.read(f)), (k, f)
        #          k: 0  1  2  3(splat) 4  5(dt) 6(dt back) 7(CURL) 8  9(write) 10 11
        assert adopted == [0, 1, 1, 0, 1, 0, 0, 0, 1, 0, 1, 1], adopted
    finally:
        a.close(); b.close()


@pytest.mark.gpu
def test_packed_dye_with_a_coarser_sim_grid_leaves_the_same_bits():
    """the same when the dye grid differs from the sim grid (the reference's default shape: dye 8 x sim): the dye pass alone runs on the
    packed field (k_advect_dye_fast_rgb: 24 instead of 32 B/texel) once the DYE grid has 3072^2 texels"""
    import fluid_hip
    DT = 0.016666
    cfg = {"SIM_RESOLUTION": 512, "DYE_RESOLUTION": 3072, "PRESSURE_ITERATIONS": 20}
    a = fluid_hip.FluidSim(canvas=(3072, 3072), config=cfg, schedule="passes", random=fluid_hip.mulberry32(7))
    b = fluid_hip.FluidSim(canvas=(3072, 3072), config=cfg, schedule="fused", random=fluid_hip.mulberry32(7))
    try:
        a.multipleSplats(5); b.multipleSplats(5)
        assert b.schedule_info(20)["dye_packed"] == 1
        a.step(DT, 20); b.step(DT, 20)
        for f in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(a.read(f), b.read(f)), f
        for s_ in (a, b):
            s_.splat(0.6, 0.2, 100.0, 300.0, {"r": 0.1, "g": 0.1, "b": 0.4})
        a.step(DT, 17); b.step(DT, 17)
        assert np.array_equal(a.read("dye"), b.read("dye")) and np.array_equal(a.read("velocity"), b.read("velocity"))
        img_a, img_b = a.render(256, 256), b.render(256, 256)       # the compositor reads RGBA: unpacks
        assert np.array_equal(img_a, img_b)
    finally:
        a.close(); b.close()


@pytest.mark.gpu
def test_packed_dye_leaves_the_same_bits():
    """At and above 3072^2 texels the fused advection runs on the dye packed to three floats per texel while the context knows its alpha
    to be one value (1 after a splat, divided by the decay at every advection); everything that reads or writes dye texels sees RGBA.
    Against the per-pass schedule, bit for bit, through: a long call (packed), reads (unpack), a splat on the packed field, a short
    read-every-few-steps pattern (the hold-off: packing stops paying), and a dye written with NON-uniform alpha (must never be packed)."""
    import fluid_hip
    DT, N = 0.016666, 3072
    a, b = sim_of(N, "passes", iters=20), sim_of(N, "fused", iters=20)
    def same(tag):
        for f in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(a.read(f), b.read(f)), (tag, f)
    try:
        a.multipleSplats(6); b.multipleSplats(6)
        assert b.schedule_info(20)["dye_packed"] == 1 and a.schedule_info(20)["dye_packed"] == 0
        a.step(DT, 20); b.step(DT, 20)
        same("20 steps packed")                      # alpha = 1 / decay^20 comes back from the scalar
        assert b.schedule_info(1)["dye_packed"] == 1  # 20 advections between two readers: packing keeps paying
        b.step(DT, 1); a.step(DT, 1)                  # packs again ...
        for s_ in (a, b):
            s_.splat(0.3, 0.7, -400.0, 250.0, {"r": 0.2, "g": 0.5, "b": 0.1})   # ... and the splat lands on the PACKED field (alpha -> 1)
        a.step(DT, 2); b.step(DT, 2)
        same("splat on the packed field")             # a reader after 3 advections: hold-off
        assert b.schedule_info(1)["dye_packed"] == 0
        a.step(DT, 2); b.step(DT, 2)
        same("held off: RGBA")
        d = b.read("dye")
        d[..., 3] = np.linspace(0.25, 2.0, d.shape[0] * d.shape[1], dtype=np.float32).reshape(d.shape[:2])   # alpha is data now
        a.write("dye", d); b.write("dye", d)
        a.step(DT, 3); b.step(DT, 3)
        same("non-uniform alpha is advected, not replaced by a scalar")
        assert b.schedule_info(400)["dye_packed"] == 0
    finally:
        a.close(); b.close()


@pytest.mark.gpu
def test_packed_dye_is_not_tried_where_its_kernel_does_not_apply():
    """a dye decay outside [1, 2) (DENSITY_DISSIPATION 100: 1 + 100 dt = 2.67) takes the general advection kernel, which reads RGBA: the field
    must not be packed for it (and the bits must be the per-pass schedule's either way)"""
    a, b = sim_of(3072, "passes", iters=10, DENSITY_DISSIPATION=100), sim_of(3072, "fused", iters=10, DENSITY_DISSIPATION=100)
    try:
        a.multipleSplats(4); b.multipleSplats(4)
        a.step(0.016666, 3); b.step(0.016666, 3)
        for f in ("velocity", "dye"):
            assert np.array_equal(a.read(f), b.read(f)), f
        b.config["DENSITY_DISSIPATION"] = a.config["DENSITY_DISSIPATION"] = 1
        a.step(0.016666, 2); b.step(0.016666, 2)      # now it applies: packs (no hold-off: the field was never packed before)
        assert np.array_equal(a.read("dye"), b.read("dye"))
    finally:
        a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sim_res,dye_res,seed", [(700, 700, 1), (700, 700, 11), (1100, 1100, 2), (3200, 3200, 3), (3200, 3200, 13), (96, 700, 4), (96, 700, 14),
                                                  (200, 3100, 5), (2100, 2100, 6)])
def test_random_call_sequences_leave_what_the_passes_leave(sim_res, dye_res, seed):
    """The library keeps state a caller cannot see — the next step's curl / vorticity / divergence computed ahead, the dye packed to three
    floats with its alpha as a scalar, the hold-off counters — and every entry point has to keep it honest.  A seeded random sequence of
    calls (steps of 1 / 2 / 5, splats, reads, writes of single fields, dt and CURL changes, a render, a raw pointer) on the fused
    schedule against the same sequence on the per-pass schedule: all five fields bit for bit at every read and at the end.  Sizes on both
    sides of every threshold: chained + working ahead (700), chained without (1100 … is below 1536: with), packed dye (3200), dye != sim
    chained (96 / 700), dye != sim with the dye packed (200 / 3100)."""
    import fluid_hip
    rng = np.random.default_rng(seed)
    side = max(sim_res, dye_res)
    mk = lambda sched: fluid_hip.FluidSim(canvas=(side, side), config={"SIM_RESOLUTION": sim_res, "DYE_RESOLUTION": dye_res, "PRESSURE_ITERATIONS": 20},
                                          schedule=sched, random=fluid_hip.mulberry32(99))
    a, b = mk("passes"), mk("fused")
    fields = ("velocity", "pressure", "divergence", "curl", "dye")
    dt = 0.016666
    try:
        a.multipleSplats(4); b.multipleSplats(4)
        for op_i in range(90):
            op = rng.integers(0, 10)
            if op <= 3:
                n = int(rng.choice([1, 1, 2, 5]))
                a.step(dt, n); b.step(dt, n)
            elif op == 4:
                x, y = float(rng.random()), float(rng.random())
                col = {"r": float(rng.random()), "g": float(rng.random()), "b": float(rng.random())}
                for s_ in (a, b):
                    s_.splat(x, y, 500.0 * (x - 0.5), -300.0 * (y - 0.5), col)
            elif op == 5:
                f = fields[int(rng.integers(0, 5))]
                assert np.array_equal(a.read(f), b.read(f)), (op_i, "read", f)
            elif op == 6:
                f = ("velocity", "dye", "pressure")[int(rng.integers(0, 3))]
                v = b.read(f) * np.float32(0.75)
                if f == "dye" and rng.random() < 0.5:
                    v[..., 3] = np.float32(rng.random())          # a dye with ANOTHER uniform alpha: still one value
                a.write(f, v); b.write(f, v)
            elif op == 7:
                dt = float(rng.choice([0.016666, 0.01, 0.016666]))
            elif op == 8:
                c_ = int(rng.choice([30, 10, 0]))
                a.config["CURL"] = b.config["CURL"] = c_
            else:
                if rng.random() < 0.5:
                    assert np.array_equal(a.render(128, 128), b.render(128, 128)), (op_i, "render")
                else:
                    b.device_view("dye")                           # a raw pointer: the library forgets what it knew about the dye
        for f in fields:
            assert np.array_equal(a.read(f), b.read(f)), ("end", f)
    finally:
        a.close(); b.close()
