"""The in-tree libraries are what the sources say: `make -q` (question mode) finds nothing to rebuild, for the product and for the lab build.
Round 5, visit 5 measured two kernel variants on STALE libraries — a source that no longer compiled on the host side, a build step whose
output nobody read, and a GPU visit that happily loaded the previous .so (profiles/r05/dye_ne_sim_lds_ab.txt).  The libraries travel to the
GPU box prebuilt, so this is the place to catch it: the CPU suite, in front of every visit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "webgl-fluid-simulation_amd")


@pytest.mark.parametrize("flavor", [[], ["PROBES=1"]])
def test_the_library_is_up_to_date_with_its_sources(flavor):
    lib = os.path.join(PKG, "libfluid_hip_probes.so" if flavor else "libfluid_hip.so")
    if not os.path.exists(lib):
        pytest.skip("%s is not built here (python -c 'import __graft_entry__ as g; g.build()')" % os.path.basename(lib))
    r = subprocess.run(["make", "-q", "-C", PKG] + flavor, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        # stale: rebuild NOW and fail only if that does not work — a checkout that touched a source's mtime must not turn the suite red,
        # a source that does not compile must
        b = subprocess.run(["make", "-C", PKG, "-j4"] + flavor, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert b.returncode == 0, "%s was older than its sources and does NOT rebuild:\n%s" % (os.path.basename(lib), b.stdout.decode(errors="replace")[-1500:])
