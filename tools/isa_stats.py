#!/usr/bin/env python3
"""Static per-kernel statistics of the gfx950 device code: instruction counts by unit and register use.
Usage: tools/isa_stats.py [name-fragment ...]   (compiles csrc/fluid_kernels.hip to assembly under /tmp, device side only)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "webgl-fluid-simulation_amd", "csrc", "fluid_kernels.hip")


def main():
    out = "/tmp/fluid_kernels.s"
    extra = os.environ.get("EXTRA", "").split()
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S",
                    "--cuda-device-only", "-o", out, SRC] + extra, check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    frags = sys.argv[1:]
    for m in re.finditer(r"^(_Z\S+):\s*; @\S+\n(.*?)\.Lfunc_end\d+:", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", dem.replace("fluid::(anonymous namespace)::", "").replace("void ", ""))
        if frags and not any(f in short for f in frags):
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            t = line.strip()
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            op = t.split()[0]
            unit = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                    "vmem_ld" if re.match(r"(global|buffer|flat)_load", op) else "vmem_st" if re.match(r"(global|buffer|flat)_(store|atomic)", op) else
                    "lds" if op.startswith("ds_") else "other")
            c[unit] += 1
            if "dpp" in t:
                c["dpp"] += 1
        regs = {k: re.search(r"\.set %s\.%s, (\d+)" % (re.escape(name), k), s) for k in ("num_vgpr", "numbered_sgpr", "private_seg_size")}
        print("%-46s %s  %s" % (short[:46], " ".join("%s=%d" % kv for kv in sorted(c.items())),
                                " ".join("%s=%s" % (k, v.group(1)) for k, v in regs.items() if v)))


if __name__ == "__main__":
    main()
