"""Stated parity tolerances (relative to max|field|) and — since round 2 — the list of fixtures that are held BIT FOR BIT.

The reference pipeline is bit-reproducible: every pass is plain fp32 arithmetic in a fixed order, sqrt and divide are correctly
rounded, and its one transcendental — exp() in splatShader (script.js:738) — is a fixed polynomial in the rasteriser the reference
runs on (SwiftShader's exponential2, restated in oracle/fluid_oracle.c and csrc/fluid_math.h).  So at POWER-OF-TWO grid sizes, where the
texel-centre coordinates (i + .5) / W are exact in fp32 and the rasteriser-interpolated coordinates carry no jitter, whole runs —
splats, 50 steps with CURL = 30, the 4096^2 headline workload — reproduce bit for bit.  At other sizes the live reference samples its
LINEAR-filtered velocity / dye textures at interpolated coordinates that are an ulp off the formula, which leaks ~W * 2^-22 of the
neighbour difference into every tap ("texcoord jitter"): those fixtures keep tolerances.  Passes that only read NEAREST textures (clear,
Jacobi) are bit-reproducible at every size.
"""

BITWISE_PASSES = ("clear", "jacobi", "jacobi5")  # golden name fragments gated with array_equal on pressure

# fixtures whose sim AND dye grids are powers of two: every field is held with array_equal (oracle and HIP, both schedules)
BITWISE_FIXTURES = ("splats_only_64", "splat_stream_20", "step1_64", "step5_curl0_64", "step10_64", "step5_sim32_dye128")
# sim grid a power of two, dye grid not (64 x 32 / 96 x 48): the four sim-grid fields are bitwise, the dye keeps its tolerance
BITWISE_SIM_FIELDS = ("step3_wide_64x32_dye96x48",)


def bitwise_fields(name: str):
    """the fields of a golden that must be bit-identical to the reference"""
    if name in BITWISE_FIXTURES:
        return ("velocity", "pressure", "divergence", "curl", "dye")
    if name in BITWISE_SIM_FIELDS:
        return ("velocity", "pressure", "divergence", "curl")
    return ()


def golden_tolerance(name: str) -> float:
    if name.startswith("pass_"):
        if "_noise_" in name:
            return 4e-5      # white-noise inputs: jitter leak scales with roughness (measured <= 2.0e-5)
        return 8e-6          # smooth inputs (measured <= 3.4e-6)
    table = {
        "splats_only_64": 0.0,                     # bitwise (BITWISE_FIXTURES)
        "splat_stream_20": 0.0,
        "step1_64": 0.0,
        "step2_params_48": 2e-5,                   # measured 6.0e-6
        "step3_wide_64x32_dye96x48": 2e-5,         # the dye grid (96 x 48) only: measured 5.9e-6; the sim-grid fields are bitwise
        "step3_tall_24x60": 1.5e-4,                # measured 3.2e-5
        "step5_curl0_64": 0.0,
        "step5_sim32_dye128": 0.0,
        "step10_64": 0.0,                          # ten steps with CURL = 30: bitwise (it is the exp() of the splats that used to differ)
        "resize_32_to_48_dye64": 2e-5,             # measured 2.7e-6
    }
    return table[name]


# HIP vs CPU oracle on identical inputs (both restate the same arithmetic; sqrt and divide are correctly rounded on both sides and
# exp is the same polynomial): every pass is expected bitwise; the allowance below is kept for the sqrt / divide passes only in case a
# toolchain ever relaxes them.
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
