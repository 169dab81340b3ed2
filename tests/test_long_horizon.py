"""Long horizons against the live reference.  50 steps with CURL = 0: the trajectory stays comparable
texel by texel (tolerance 1e-4 of max|field|).  50 steps with CURL = 30: the vorticity force f / (|f| + 1e-4) is
discontinuous where grad|curl| ~ 0 (script.js:856-857), so rounding differences are amplified and individual texels
decorrelate — what remains comparable are the statistics: kinetic energy, total dye, peak speed (within 2 %; the
reference's own run-to-run variation is zero, the band covers the chaotic sensitivity measured in the survey)."""
import numpy as np
import pytest

import scenario as S

CURL0_TOL = 1e-4
STAT_TOL = 0.02


def stats(f):
    v = f["velocity"].astype(np.float64)
    return {"kinetic": 0.5 * float((v ** 2).sum()), "dye": float(f["dye"][..., :3].astype(np.float64).sum()), "vmax": float(np.abs(v).max())}


def check(out, log, g, name):
    assert np.array_equal(log, g["splats"])
    ref = {k: g["out_" + k] for k in S.FIELDS}
    if "curl0" in name:
        for k in S.FIELDS:
            assert S.rel_err(out[k], ref[k]) <= CURL0_TOL, (k, S.rel_err(out[k], ref[k]))
    a, b = stats(out), stats(ref)
    for k in a:
        assert abs(a[k] - b[k]) <= STAT_TOL * abs(b[k]), (k, a[k], b[k])
    assert np.isfinite(out["velocity"]).all() and np.abs(out["velocity"]).max() <= 1000.0 * 1.05


@pytest.mark.parametrize("name", ["long50_curl0_64", "long50_curl30_64"])
def test_oracle_long_horizon(oracle, name):
    g, sc = S.load(name)
    ad = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    out, log = S.replay(ad, g, sc)
    check(out, log, g, name)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["fused", "passes"])
@pytest.mark.parametrize("name", ["long50_curl0_64", "long50_curl30_64"])
def test_hip_long_horizon(name, schedule):
    g, sc = S.load(name)
    ad = S.HipAdapter(S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), schedule=schedule)
    try:
        out, log = S.replay(ad, g, sc)
    finally:
        ad.close()
    check(out, log, g, name)


# ---- BASELINE configs[1] size against the live reference: 1024^2, 50 iterations, 2 steps (subsampled + one full band) ------
# Relative to max|field|.  The live reference samples its LINEAR velocity / dye textures at rasteriser-interpolated fp32
# coordinates; the leak of neighbour differences into every tap grows with the width (~ W * 2^-22), and
# curl / divergence are differences of those taps, so the reference's own noise floor at W = 1024 is ~1e-4 … 1e-3 of max|field|
# for them (measured restatement-vs-reference: velocity 5.5e-5, pressure 2.7e-5, divergence 6.8e-4, curl 1.2e-3, dye 5.9e-6).
# The HIP path is held to the restatement far tighter (bitwise / 3e-5: tests/test_hip_vs_oracle.py).
BIG_TOL = {"velocity": 2e-4, "pressure": 1e-4, "divergence": 2e-3, "curl": 3.5e-3, "dye": 2e-5}


def _check_big(out, log, g):
    assert np.array_equal(log, g["splats"])
    b0, b1 = (int(x) for x in g["band"])
    for k in S.FIELDS:
        scale = float(g["absmax_" + k])
        a = out[k]
        assert float(np.abs(a[::8, ::8].astype(np.float64) - g["sub8_" + k]).max()) <= BIG_TOL[k] * scale, k
        assert float(np.abs(a[b0:b1].astype(np.float64) - g["band_" + k]).max()) <= BIG_TOL[k] * scale, k
        assert abs(float(np.abs(a).max()) - scale) <= BIG_TOL[k] * scale, k


def test_oracle_matches_live_reference_at_1024(oracle):
    g, sc = S.load("big_step2_1024")
    ad = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    out, log = S.replay(ad, g, sc)
    _check_big(out, log, g)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["fused", "passes"])
def test_hip_matches_live_reference_at_1024(schedule):
    g, sc = S.load("big_step2_1024")
    ad = S.HipAdapter(S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), schedule=schedule)
    try:
        out, log = S.replay(ad, g, sc)
    finally:
        ad.close()
    _check_big(out, log, g)


# ---- the HEADLINE size against the live reference: BASELINE configs[2], 4096^2, 50 iterations, bench.py's 20 seeded splats, 2 steps
# (oracle/live/make_golden_4096.py: every 32nd row / column + one full band of 8 rows, sampled inside the page).  The reference's own
# noise floor grows with the width (texcoord jitter ~ W * 2^-22 = 1e-3 at W = 4096); measured
# restatement-vs-reference: velocity 2.5e-3, pressure 5.0e-4, divergence 6.1e-3, curl 5.1e-3, dye 4.2e-5 of max|field|; the splat list
# and max|pressure| agree exactly.  The HIP path is additionally held to the restatement at this size (median / 99th percentile / max).
# (curl: 1.3e-2 on the 4096 x 8192 grid, where max|curl| is half as large)
HUGE_TOL = {"velocity": 5e-3, "pressure": 1.2e-3, "divergence": 1.5e-2, "curl": 2.5e-2, "dye": 1e-4}


def _check_huge(out, log, g):
    assert np.array_equal(log, g["splats"])
    st = int(g["stride"])
    b0, b1 = (int(x) for x in g["band"])
    for k in S.FIELDS:
        scale = float(g["absmax_" + k])
        a = out[k]
        assert a.shape[:2] == (int(g["sim"][1]), int(g["sim"][0]))
        assert float(np.abs(a[::st, ::st].astype(np.float64) - g["sub_" + k]).max()) <= HUGE_TOL[k] * scale, k
        assert float(np.abs(a[b0:b1].astype(np.float64) - g["band_" + k]).max()) <= HUGE_TOL[k] * scale, k
        assert abs(float(np.abs(a).max()) - scale) <= HUGE_TOL[k] * scale, k


def test_oracle_matches_live_reference_at_4096(oracle):
    g, sc = S.load("big_step2_4096")
    ad = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    out, log = S.replay(ad, g, sc)
    _check_huge(out, log, g)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["fused", "passes"])
def test_hip_matches_live_reference_and_oracle_at_4096(oracle, schedule):
    g, sc = S.load("big_step2_4096")
    ad = S.HipAdapter(S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), schedule=schedule)
    try:
        out, log = S.replay(ad, g, sc)
    finally:
        ad.close()
    _check_huge(out, log, g)
    ref = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    want, _ = S.replay(ref, g, sc)
    # HIP vs the restatement at this size: identical for almost every texel, heavy-tailed where the vorticity force f / (|f| + 1e-4)
    # is discontinuous (grad|curl| ~ 0, script.js:856-857): there the 1-ulp exp() difference of the splats picks another direction
    # and moves the velocity by up to CURL * |curl| * dt.  Measured (tools/hip_vs_oracle_4096.py, of max|field|): median <= 2.6e-8,
    # 99th percentile <= 1.1e-4, max 1.9e-3 (velocity) … 1.4e-2 (curl) — the same tail the restatement has against the reference.
    for k in S.FIELDS:
        d = np.abs(out[k].astype(np.float64) - want[k]).ravel() / float(np.abs(want[k]).max())
        assert float(np.median(d)) <= 1e-6, (k, float(np.median(d)))
        assert float(np.quantile(d, 0.99)) <= 3e-4, (k, float(np.quantile(d, 0.99)))
        assert float(d.max()) <= 2 * HUGE_TOL[k], (k, float(d.max()))


# ---- BASELINE configs[3]'s grid (8192^2, 50 iterations; the 4-GPU configuration) through the live reference as well
# (oracle/live/make_golden_8192.py: every 64th row / column + a band of 4 rows).  Measured restatement-vs-reference: velocity 1.4e-3,
# pressure 2.5e-4, divergence 2.6e-3, curl 5.8e-3, dye 2.4e-5 of max|field| — inside the bounds stated for 4096^2.
def test_oracle_matches_live_reference_at_8192(oracle):
    g, sc = S.load("big_step2_8192")
    ad = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    out, log = S.replay(ad, g, sc)
    _check_huge(out, log, g)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["fused", "passes"])
def test_hip_matches_live_reference_at_8192(schedule):
    g, sc = S.load("big_step2_8192")
    ad = S.HipAdapter(S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), schedule=schedule)
    try:
        out, log = S.replay(ad, g, sc)
    finally:
        ad.close()
    _check_huge(out, log, g)


@pytest.mark.gpu
@pytest.mark.parametrize("tiles_x", [1, 2])
def test_four_ranks_match_live_reference_at_8192(tiles_x):
    """configs[3] as a 4-GPU run decomposes it — four row stripes, or 2 x 2 tiles (BASELINE's wording) — halo 56, the native plan,
    against the reference itself"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    g, sc = S.load("big_step2_8192")
    grp = StripeGroup(4, canvas=S.canvas_of(g), config=sc["config"], halo=56, random=fluid_hip.mulberry32(sc["seed"]), reach=32, tiles_x=tiles_x)
    try:
        grp.multipleSplats(sc["randomSplats"])
        grp.step(sc.get("dt", 0.016666), sc["steps"])
        grp.check_halo()
        out = {k: grp.read(k) for k in S.FIELDS}
    finally:
        grp.close()
    _check_huge(out, g["splats"], g)


# ---- the grid bench.py runs on TWO GPUs (weak scaling: 4096 x 8192, one 4096 x 4096 stripe per rank) through the live reference
# (oracle/live/make_golden_4096x8192.py; the full-resolution band straddles the border between the two stripes)
def test_oracle_matches_live_reference_on_the_two_gpu_bench_grid(oracle):
    g, sc = S.load("big_step2_4096x8192")
    ad = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    out, log = S.replay(ad, g, sc)
    _check_huge(out, log, g)


@pytest.mark.gpu
def test_two_stripes_match_live_reference_on_the_two_gpu_bench_grid():
    """what `bench.py --gpus 2` computes (two row stripes, halo 56, reach 32, native plan, fused schedule) against the reference"""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    g, sc = S.load("big_step2_4096x8192")
    grp = StripeGroup(2, canvas=S.canvas_of(g), config=sc["config"], halo=56, random=fluid_hip.mulberry32(sc["seed"]), reach=32)
    try:
        grp.multipleSplats(sc["randomSplats"])
        grp.step(sc.get("dt", 0.016666), sc["steps"])
        grp.check_halo()
        out = {k: grp.read(k) for k in S.FIELDS}
    finally:
        grp.close()
    _check_huge(out, g["splats"], g)


# ---- the headline grid over a longer horizon: 4096^2, 50 iterations, CURL = 0, TEN steps (500 Jacobi iterations, 20 advections) through
# the live reference (oracle/live/make_golden_4096_curl0.py).  Without the discontinuous vorticity force the trajectory is not chaotic
# and the comparison stays texel-tight — which also shows that the 1e-3 differences of the CURL = 30 fixtures above come from that
# force, not from the grid size.  Measured restatement-vs-reference after 10 steps: velocity 9.3e-7, pressure 2.2e-7, divergence 1.1e-5,
# curl 1.5e-4, dye 5.0e-7 of max|field|.
CURL0_4096_TOL = {"velocity": 4e-6, "pressure": 1e-6, "divergence": 5e-5, "curl": 6e-4, "dye": 2e-6}


def _check_curl0(out, log, g):
    assert np.array_equal(log, g["splats"])
    st = int(g["stride"])
    b0, b1 = (int(x) for x in g["band"])
    for k in S.FIELDS:
        scale = float(g["absmax_" + k])
        a = out[k]
        assert float(np.abs(a[::st, ::st].astype(np.float64) - g["sub_" + k]).max()) <= CURL0_4096_TOL[k] * scale, (k, float(np.abs(a[::st, ::st].astype(np.float64) - g["sub_" + k]).max()) / scale)
        assert float(np.abs(a[b0:b1].astype(np.float64) - g["band_" + k]).max()) <= CURL0_4096_TOL[k] * scale, k
        assert abs(float(np.abs(a).max()) - scale) <= CURL0_4096_TOL[k] * scale, k


def test_oracle_matches_live_reference_over_ten_steps_at_4096(oracle):
    g, sc = S.load("big_step10_curl0_4096")
    ad = S.OracleAdapter(oracle, S.canvas_of(g), sc.get("config"), sc.get("seed", 1234))
    out, log = S.replay(ad, g, sc)
    _check_curl0(out, log, g)


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["fused", "passes"])
def test_hip_matches_live_reference_over_ten_steps_at_4096(schedule):
    g, sc = S.load("big_step10_curl0_4096")
    ad = S.HipAdapter(S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), schedule=schedule)
    try:
        out, log = S.replay(ad, g, sc)
    finally:
        ad.close()
    _check_curl0(out, log, g)
