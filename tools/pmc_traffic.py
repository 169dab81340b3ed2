#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE
runs, as MI355X_MICROARCH.md's HBM section prescribes).

Corrections applied (gfx950, this rocprofv3):
  * both counters are in KiB (value * 1024 = bytes);
  * FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads -> x2 (the guide's gfx950 rule);
  * WRITE_SIZE is "uncalibrated" per the guide -> calibrated here on kernels of the per-pass schedule whose
    traffic is known exactly and has no reuse: k_clear (reads 4 B, writes 4 B per texel) and k_jacobi
    (reads 8 B, writes 4 B per texel); the measured/known ratios are printed and the write factor applied.

Usage: tools/pmc_traffic.py <fetch_fused.csv> <write_fused.csv> <fetch_passes.csv> <write_passes.csv> <W> <H> [out.json]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("fluid::(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name)


def per_kernel(path):
    """average counter value per dispatch (a dispatch's value = sum over its rows: one row per XCD/instance)"""
    per_dispatch = defaultdict(float)
    names = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            d = row["Dispatch_Id"]
            per_dispatch[d] += float(row["Counter_Value"])
            names[d] = short(row["Kernel_Name"])
    agg = defaultdict(list)
    for d, v in per_dispatch.items():
        agg[names[d]].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def per_kernel_counters(path):
    """{kernel: {counter: (average value per dispatch, dispatches)}} for a CSV that holds one or several counters"""
    per_dispatch = defaultdict(float)
    names = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
    agg = defaultdict(lambda: defaultdict(list))
    for (d, c), v in per_dispatch.items():
        agg[names[d]][c].append(v)
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in agg.items()}


def main(a):
    ff, wf, fp, wp = (per_kernel(p) for p in a[:4])
    W, H = int(a[4]), int(a[5])
    cells = W * H
    KiB = 1024.0
    # calibration on known-traffic kernels of the per-pass schedule
    calib = {}
    for k, rd, wr in (("k_clear", 4, 4), ("k_jacobi", 8, 4), ("k_gradsub", 12, 8)):
        if k in fp and k in wp:
            calib[k] = {"read_known_MB": rd * cells / 1e6, "FETCH_SIZE_x1024_MB": fp[k][0] * KiB / 1e6,
                        "write_known_MB": wr * cells / 1e6, "WRITE_SIZE_x1024_MB": wp[k][0] * KiB / 1e6}
            calib[k]["fetch_ratio"] = calib[k]["FETCH_SIZE_x1024_MB"] / calib[k]["read_known_MB"]
            calib[k]["write_ratio"] = calib[k]["WRITE_SIZE_x1024_MB"] / calib[k]["write_known_MB"]
    fetch_fix = 2.0
    write_fix = 1.0 / calib["k_clear"]["write_ratio"] if "k_clear" in calib else 1.0
    out = {"grid": [W, H], "units": "bytes per launch (average over dispatches)", "fetch_correction": fetch_fix,
           "write_correction": round(write_fix, 4), "calibration": calib, "kernels": {}}
    for k in sorted(set(ff) | set(wf)):
        if k not in ff or k not in wf:
            continue
        rd = ff[k][0] * KiB * fetch_fix
        wr = wf[k][0] * KiB * write_fix
        out["kernels"][k] = {"read_bytes": int(rd), "write_bytes": int(wr), "total_bytes": int(rd + wr),
                             "bytes_per_texel": round((rd + wr) / cells, 2), "dispatches": ff[k][1]}
    jt = [k for k in out["kernels"] if k.startswith("k_jacobi_tb")]
    if jt:
        out["dominant_kernel"] = jt[0]
        out["bytes_per_launch"] = out["kernels"][jt[0]]["total_bytes"]
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(a) > 6:
        open(a[6], "w").write(txt)


if __name__ == "__main__":
    main(sys.argv[1:])
