"""The bookkeeping of the persistent chained pressure loop (csrc/fluid_pchain.h; k_jacobi_pchain runs the 50 iterations of script.js:1259-1266
as one launch of persistent workgroups that take stacks of tiles from per-XCD ticket heads), checked on the host against the real header:
tickets map onto items one to one, everything an item waits for lies in an earlier band and is among the cells / bands the kernel polls, the
tiles of a stack store every row once with their aprons intact — and the protocol itself, simulated under arbitrary workgroup placement and
interleaving, runs every item once, after its dependencies, never spins on an item nobody holds, and ends.  No GPU."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "webgl-fluid-simulation_amd", "csrc")


def test_tickets_dependencies_stacks_and_the_simulated_protocol(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "pchain_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(HERE, "pchain_check.cpp")],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok:"), r.stdout[-2000:]
