"""Stated parity tolerances (relative to max|field|), from SURVEY.md Appendix C and the measured
oracle-vs-live-reference errors (oracle/live/make_golden.py run).

Why not bitwise everywhere: the live reference samples its LINEAR-filtered velocity/dye textures at
rasteriser-interpolated fp32 coordinates, which leaks ~W*2^-22 of the neighbour difference into
every tap ("texcoord jitter"), and its exp/sqrt/divide are SwiftShader's.  Passes that only read
NEAREST textures (clear, Jacobi) ARE bit-reproducible and are gated bitwise.
"""

BITWISE_PASSES = ("clear", "jacobi", "jacobi5")  # golden name fragments gated with array_equal on pressure


def golden_tolerance(name: str) -> float:
    if name.startswith("pass_"):
        if "_noise_" in name:
            return 4e-5      # white-noise inputs: jitter leak scales with roughness (measured <= 2.0e-5)
        return 8e-6          # smooth inputs (measured <= 3.4e-6)
    table = {
        "splats_only_64": 1e-6,                    # measured 2.4e-7
        "splat_stream_20": 1e-6,                   # measured 1.2e-7
        "step1_64": 5e-6,                          # measured 1.4e-6
        "step2_params_48": 2e-5,                   # measured 6.0e-6
        "step3_wide_64x32_dye96x48": 1.5e-4,       # fast edge splat, CURL=30 (measured 3.9e-5)
        "step3_tall_24x60": 1.5e-4,                # measured 3.2e-5
        "step5_curl0_64": 1e-5,                    # CURL=0 keeps the trajectory non-chaotic (measured 3.1e-6)
        "step5_sim32_dye128": 3e-5,                # measured 6.0e-6
        "step10_64": 1e-3,                         # CURL=30 trajectories decorrelate (Appendix C): 10 steps <= 1e-3
        "resize_32_to_48_dye64": 2e-5,             # measured 2.7e-6
    }
    return table[name]


# HIP vs CPU oracle on identical inputs (both restate the same arithmetic; only libm differs):
#   Jacobi / clear / gradient subtract / curl / divergence: bitwise
#   vorticity (sqrt, divide), advection (divide), splat (exp): a few ulp
HIP_VS_ORACLE_ULP_PASSES = 4e-7
# one full step, relative to max|field|.  With CURL = 30 the vorticity force f/(|f|+1e-4) is discontinuous
# where grad|curl| ~ 0 (script.js:856-857), so the 1-ulp exp() difference between ocml and glibc in the splats
# is amplified locally (measured <= 7.7e-6 at 1024^2); with CURL = 0 the step agrees to ~1e-7.
HIP_VS_ORACLE_STEP = 3e-5

# input replay (tests/test_input_replay.py): 16 frames, 14 of them stepped, CURL = 30 -> the 10-step regime above
INPUT_REPLAY = 1e-3

# fp16-storage mode (tests/test_hip_f16.py), HIP vs the oracle's fp16 mode.  A single pass: identical, or one fp16 ulp
# apart on the few texels whose fp32 results (a libm ulp apart) straddle an fp16 rounding boundary.
# Measured (tools/f16_measure.py on an MI355X, profiles/r01/f16_parity_and_speed.txt): NO flips at all — the fp16 rounding
# absorbs the libm ulps, every pass and every 3-step run below is bit-equal to the oracle.  The allowances stay for seeds
# that do land on a boundary.
F16_FLIP_FRACTION = 1e-3
# three steps, relative to max|field|: one flipped fp16 ulp is 2^-11 = 4.9e-4 of a value, and CURL = 30 amplifies it
F16_STEP_CURL0 = 1e-3
F16_STEP = 5e-3
