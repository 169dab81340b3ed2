"""The texcoord jitter, accounted for bit for bit (oracle/raster.py).

tests/tolerances.py lets the restatement and the HIP path differ from the live reference by 2e-6 ... 1.5e-4 at grid sizes that are not
powers of two and names the cause: the reference's fragment shaders do not see the texel centres (i + .5) / W but the varyings its
rasteriser interpolates, an ulp or two away, and every LINEAR fetch then leaks a little of the neighbouring texel.  This file proves that
this is the WHOLE difference.  oracle/raster.py restates the rasteriser's interpolation (plane-equation setup and per-pixel evaluation,
identified from the reference's own output) and runs the same passes on those coordinates through a sampler:

  * the interpolation restated == the varyings the live reference's own vertex shader produced, at eleven grid sizes;
  * with it EVERY golden fixture — all single passes at 40^2 and 48 x 24, the multi-step runs at 48^2, 24 x 60, 64 x 32 / 96 x 48, the
    resize, the fp16-storage fixtures — is array_equal to the live reference, not just the power-of-two ones;
  * at power-of-two sizes the interpolated coordinates ARE the texel centres and the two evaluations coincide (which is why the plain
    restatement and the HIP kernels, which read the shader text, are bit-identical to the reference there)."""
import os

import numpy as np
import pytest

import scenario as S

from oracle import raster  # noqa: E402

f32 = np.float32
COORDS = np.load(os.path.join(S.GOLDEN_DIR, "raster_varyings.npz"))
SIZES = [tuple(int(v) for v in s) for s in COORDS["sizes"]]
NAMES = S.golden_names() + S.f16_golden_names()


@pytest.mark.parametrize("W,H", SIZES)
def test_interpolated_varyings_match_the_live_rasteriser(W, H):
    V = raster.Varyings(W, H, f32(1.0 / W), f32(1.0 / H))
    key = "%dx%d_" % (W, H)
    for name, got in (("uv_x", V.ux[0]), ("uv_y", V.uy[:, 0]), ("l_x", V.lx[0]), ("r_x", V.rx[0]), ("t_y", V.ty[:, 0]), ("b_y", V.by[:, 0])):
        assert np.array_equal(got, COORDS[key + name]), (W, H, name)
    centres = (np.arange(W, dtype=f32) + f32(0.5)) / f32(W)
    pow2 = (W & (W - 1)) == 0 and (H & (H - 1)) == 0
    assert np.array_equal(V.ux[0], centres) == pow2        # texel centres exactly at power-of-two sizes, and only there (of these sizes)


def test_fixture_list():
    assert len(NAMES) == 120 and sum(1 for n in NAMES if not n.endswith(("_64", "_128x64")) and "64" not in n and "256" not in n and "128" not in n) >= 40


@pytest.mark.parametrize("name", NAMES)
def test_every_fixture_is_bit_identical_on_the_rasterisers_coordinates(name):
    g, sc = S.load(name)
    ad = S.OracleAdapter(raster._api(), S.canvas_of(g), sc.get("config"), sc.get("seed", 1234), storage="f16" if name.startswith("f16_") else "f32")
    out, log = S.replay(ad, g, sc)
    assert np.array_equal(log, g["splats"])
    for k in S.FIELDS:
        assert np.array_equal(out[k], g["out_" + k]), (name, k, S.rel_err(out[k], g["out_" + k]))


@pytest.mark.parametrize("W,H", [(64, 64), (128, 32), (16, 256)])
def test_both_evaluations_coincide_at_power_of_two_sizes(oracle, W, H):
    rng = np.random.default_rng(W + H)
    vel = rng.normal(0, 40, (H, W, 2)).astype(f32)
    crl = rng.normal(0, 30, (H, W)).astype(f32)
    prs = rng.normal(0, 30, (H, W)).astype(f32)
    div = rng.normal(0, 30, (H, W)).astype(f32)
    dye = np.abs(rng.normal(0, 1, (H, W, 4))).astype(f32)
    dt = f32(0.016666)
    assert np.array_equal(raster.curl(vel), oracle.curl(vel))
    assert np.array_equal(raster.vorticity(vel, crl, 30.0, dt), oracle.vorticity(vel, crl, f32(30.0), dt))
    assert np.array_equal(raster.divergence(vel), oracle.divergence(vel))
    assert np.array_equal(raster.clear(prs, 0.8), oracle.clear(prs, f32(0.8)))
    assert np.array_equal(raster.jacobi(prs, div), oracle.jacobi(prs, div))
    assert np.array_equal(raster.gradsub(prs, vel), oracle.gradsub(prs, vel))
    assert np.array_equal(raster.advect(vel, vel, dt, 0.2), oracle.advect(vel, vel, dt, f32(0.2)))
    assert np.array_equal(raster.advect(vel, dye, dt, 1.0), oracle.advect(vel, dye, dt, f32(1.0)))
    assert np.array_equal(raster.splat(dye, 0.3, 0.6, W / H, 0.0025 * max(W / H, 1.0), (0.9, 0.2, 0.4)),
                          oracle.splat(dye, f32(0.3), f32(0.6), f32(W / H), f32(0.0025 * max(W / H, 1.0)), (f32(0.9), f32(0.2), f32(0.4))))


@pytest.mark.parametrize("W,H", [(40, 40), (250, 130)])
def test_the_jitter_is_as_small_as_the_tolerances_say(oracle, W, H):
    """one pass on smooth data: the two evaluations differ, by less than the per-pass tolerance of tests/tolerances.py"""
    from tolerances import golden_tolerance
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    vel = np.stack([300 * np.sin(2 * np.pi * x / W) * np.cos(2 * np.pi * y / H), 300 * np.cos(4 * np.pi * x / W)], -1).astype(f32)
    a, b = raster.advect(vel, vel, f32(0.016666), 0.2), oracle.advect(vel, vel, f32(0.016666), f32(0.2))
    assert not np.array_equal(a, b)
    assert S.rel_err(a, b) <= golden_tolerance("pass_advect_velocity_smooth_40") * W / 40
