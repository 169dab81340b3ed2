"""The product library's kernels, held to the builder's own rule (a kernel that spills at all costs 1.5 - 3 x): every
__global__ function that `make` puts into libfluid_hip.so compiles to NO scratch, no AGPRs and at most 128 VGPRs (four waves per SIMD at
least) — checked on the CPU by compiling the three .hip sources with the Makefile's flags and -Rpass-analysis=kernel-resource-usage
(tools/kernel_resources.py; hipcc cross-compiles gfx950 without a GPU).  The lab build (make PROBES=1) is exempt: its spilling shapes are
the measurements that kept them out of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_product_kernels_do_not_spill():
    import kernel_resources
    res = kernel_resources.resources(probes=False)
    assert len(res) >= 85, sorted(res)                      # every kernel of the three sources was seen
    names = " ".join(res)
    for must in ("k_jacobi_tb_mix<8, 10, 7, 12, 10, 2>", "k_jacobi_tb<8, 5, 12, 10, 2>", "k_jacobi_tb2<8, 5, 12, 10, 3>", "k_curl_vort_div_mix<8, 5, 3>",
                 "k_advect_both_fast<4>", "k_advect_cvd<4, 8, 3, 2, true>", "k_advect_cvd<4, 8, 3, 2, false>", "k_advect_both_fast_rgb<4>", "k_gradsub4", "k_display"):
        assert must in res, must
    # lab shapes stay out of the product
    for lab in ("k_jacobi_tb<8, 12, 12, 10, 2>", "k_jacobi_tb_mix2", "k_advect_both_fast<8>", "k_advect_cvd<16, 8, 4"):
        assert lab not in names, lab
    bad = {k: v for k, v in res.items() if v["scratch"] != 0 or v["agpr"] != 0 or v["vgpr"] > 128 or v["vgpr"] < 0}
    assert not bad, bad


def test_product_library_is_small_and_reads_no_tuning_knob():
    lib = os.path.join(ROOT, "webgl-fluid-simulation_amd", "libfluid_hip.so")
    assert os.path.exists(lib)
    assert os.path.getsize(lib) < 2 ** 20, os.path.getsize(lib)     # 0.53 MB (compressed code objects); 3.3 MB with the 137 lab instantiations of round 3
    blob = open(lib, "rb").read()
    for knob in (b"FLUID_TB_VARIANT", b"FLUID_TB_TAIL", b"FLUID_CHAIN", b"FLUID_FOLD_GRADSUB", b"FLUID_ADVECT_ROWS", b"FLUID_CVD_TAIL", b"FLUID_XCD_REMAP",
                 b"FLUID_SKIP_CURL", b"FLUID_STRIPE_OVERLAP"):
        assert knob not in blob, knob    # lab_env() compiles to nothing in the product: the strings are not even in the binary
    from fluid_hip import _abi
    assert _abi.lib().fluid_build_flavor() == b"product"
