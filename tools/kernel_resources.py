#!/usr/bin/env python3
"""Compile the library's .hip sources for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel: VGPRs, SGPRs,
occupancy (waves/SIMD), scratch, static LDS.  With the product's flags (default) these are the kernels of libfluid_hip.so; `--probes` adds
-DFLUID_PROBES (libfluid_hip_probes.so: the lab shapes, some of which spill — which is why they are not in the product).
Usage: tools/kernel_resources.py [--probes] [filter]      (tests/test_kernel_resources.py holds the product to scratch 0, <= 128 VGPRs)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "webgl-fluid-simulation_amd")
SOURCES = ("fluid_kernels.hip", "fluid_kernels_f16.hip", "fluid_display.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]   # = the Makefile's HIPFLAGS


def resources(probes=False, sources=SOURCES):
    """{kernel name: {"vgpr", "agpr", "sgpr", "occupancy", "scratch", "lds"}} of every __global__ function in `sources`"""
    out = {}
    procs = []
    for src in sources:
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + (["-DFLUID_PROBES"] if probes else []) + os.environ.get("EXTRA", "").split() + [
               "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(PKG, "csrc", src), "-o", "/dev/null"]
        procs.append(subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True))
    names = []
    for p in procs:
        err = p.communicate()[1]
        if p.returncode != 0:
            raise RuntimeError(err[-2000:])
        cur = None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
                out[cur] = {}
                names.append(cur)
                continue
            m = re.search(r"remark:\s+(\w[\w /\[\]]*?): (\S+)", line)
            if m and cur:
                out[cur][m.group(1).strip()] = m.group(2)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for k, d in zip(names, dem):
        v = out[k]
        name = re.sub(r"\(.*", "", d.replace("fluid::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
        res[name] = {"vgpr": int(v.get("VGPRs", -1)), "agpr": int(v.get("AGPRs", -1)), "sgpr": int(v.get("TotalSGPRs", -1)),
                     "occupancy": int(v.get("Occupancy [waves/SIMD]", -1)), "scratch": int(v.get("ScratchSize [bytes/lane]", -1)),
                     "lds": int(v.get("LDS Size [bytes/block]", -1))}
    return res


def main():
    args = [a for a in sys.argv[1:] if a != "--probes"]
    flt = args[0] if args else ""
    for name, v in resources(probes="--probes" in sys.argv).items():
        if flt and flt not in name:
            continue
        print("%-40s VGPR %4d AGPR %3d SGPR %4d occ %2d scratch %4d LDS %6d" % (name, v["vgpr"], v["agpr"], v["sgpr"], v["occupancy"], v["scratch"], v["lds"]))


if __name__ == "__main__":
    main()
