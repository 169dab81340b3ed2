"""fp16-storage mode of the oracle: the fp32 passes with an fp16 round trip of every output.
CPU only.  The rounding is pinned against numpy's float16 (every half value, every midpoint between neighbouring halves
and its fp32 neighbours, random bit patterns); the fused C step must equal the pass-by-pass composition."""
import numpy as np
import pytest

import scenario as S


def test_round_half_is_numpy_float16(oracle):
    rng = np.random.default_rng(0)
    h = np.arange(0, 0x7c01, dtype=np.uint16).view(np.float16).astype(np.float64)   # 0 … 65504, inf
    mids = ((h[:-1] + h[1:]) / 2).astype(np.float32)                                 # exact ties (fp32 holds them)
    near = np.concatenate([mids, np.nextafter(mids, np.float32(np.inf)), np.nextafter(mids, np.float32(-np.inf)), h.astype(np.float32)])
    x = np.concatenate([rng.integers(0, 2 ** 32, 500_000, dtype=np.uint64).astype(np.uint32).view(np.float32), near, -near,
                        np.array([65519.996, 65520.0, 65520.004, 1e30, np.inf, -np.inf, 0.0, -0.0, 2.0 ** -25, 2.0 ** -24], np.float32)])
    x = x[~np.isnan(x)]
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    got = oracle.round_half(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_step_f16_is_the_pass_by_pass_composition(oracle):
    cfg = {"SIM_RESOLUTION": 48, "DYE_RESOLUTION": 80, "PRESSURE_ITERATIONS": 9}
    a = S.OracleAdapter(oracle, (256, 192), cfg, 5, storage="f16")
    b = S.OracleAdapter(oracle, (256, 192), cfg, 5, storage="f16")
    a.multiple_splats(4); b.multiple_splats(4)
    for _ in range(2):
        a.step(0.016666, 1)
        for p in ["curl", "vorticity", "divergence", "clear"] + ["jacobi"] * 9 + ["gradsub", "advect_velocity", "advect_dye"]:
            b.run_pass(p, 0.016666)
    fa, fb = a.fields(), b.fields()
    for k in S.FIELDS:
        assert np.array_equal(fa[k], fb[k]), k
        assert np.array_equal(fa[k], fa[k].astype(np.float16).astype(np.float32)), k   # every stored value is a half


def test_f16_storage_tracks_f32_storage(oracle):
    """a sanity bound, not a parity claim: after a few steps the two storage modes still describe the same flow"""
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 20}
    a = oracle.RefSim(canvas=(256, 256), config=cfg, seed=3, storage="f16")
    b = oracle.RefSim(canvas=(256, 256), config=cfg, seed=3, storage="f32")
    a.multiple_splats(5); b.multiple_splats(5)
    a.step(0.016666, 3); b.step(0.016666, 3)
    for k in ("velocity", "dye"):
        assert S.rel_err(a.fields()[k], b.fields()[k]) < 0.1, k   # measured 0.026 (velocity), 3 steps
    with pytest.raises(ValueError):
        oracle.RefSim(storage="bf16")
