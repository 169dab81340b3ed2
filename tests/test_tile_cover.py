"""The tiling arithmetic every register-tile kernel shares (csrc/fluid_tiles.h: make_axis, tile_exact, tile_of_block), checked on the host
against the real header: exact ranges cover an output range once and only once for every tile span / apron the kernels use (the two-texel
Jacobi tile and k_advect_cvd included), keep their apron from the tile's rim, and the XCD-aware block order is a bijection.  No GPU: the
harness only calls the header's host-callable integer functions (built with hipcc because the header pulls in the HIP runtime headers)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "webgl-fluid-simulation_amd", "csrc")


def test_tiles_cover_every_range_exactly_once(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "tile_cover_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(HERE, "tile_cover_check.cpp")],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok:"), r.stdout[-2000:]
