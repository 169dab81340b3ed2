"""div_uniform (csrc/fluid_math.h): the fp32 quotient by a wave-uniform divisor as a double multiply by the divisor's reciprocal — what
the fused advection kernel (k_advect_both_fast) divides with.  The kernel itself is held to the per-pass kernels (which divide the plain
way) and to the goldens on the GPU; here the arithmetic identity is checked on the host over every class of operand: 2.5e6 texel-centre
coordinates (i + .5) / W and 2.4e8 (decay, value) pairs incl. subnormal inputs and results, signed zeros, infinities."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_double_reciprocal_multiply_equals_fp32_divide(tmp_path):
    exe = str(tmp_path / "div_uniform_check")
    subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "div_uniform_check.c"), "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.splitlines()
    assert lines[0].startswith("coordinates:") and lines[0].endswith(", 0 differ") and lines[1].startswith("decays:") and lines[1].endswith(", 0 differ"), r.stdout
