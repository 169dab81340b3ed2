"""The geometry of a pressure launch cut around an exchange in flight (csrc/fluid_cut.h: cut_depths, block_cut, cut_frame — what
fluid_solver.cpp's pass_jacobi launches while ghost texels travel and what follows), checked on the host against the real header: interior +
frame cover the band once, an interior reads nothing outside the owned rectangle, the second cut launch keeps clear of the first one's
frame inputs and of the rows in flight.  No GPU (g++)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "webgl-fluid-simulation_amd", "csrc")


def test_cut_geometry_invariants(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not found")
    exe = str(tmp_path / "cut_check")
    subprocess.run([gxx, "-O1", "-std=c++17", "-Wall", "-I", CSRC, "-o", exe, os.path.join(HERE, "cut_check.cpp")], check=True, capture_output=True,
                   timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok:"), r.stdout[-2000:]
