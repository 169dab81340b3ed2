"""DOCS LINT, not a test of bench.py (that is tests/test_bench_live.py, which runs it): the bench lines COMMITTED under profiles/ — the
numbers DESIGN.md quotes — are well-formed lines of the contract (profiles/r03/bench_4096_50.json is what `python bench.py` printed there): the keys the driver reads, a roofline that is a fraction of a physical peak (<= 1, = achieved / peak,
achieved = measured HBM bytes per launch / measured launch time), the CPU baseline of the same run and what kind it is."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r06"   # the CURRENT round's files (the lint still read r03)
FILES = ["bench_4096_50.json", "bench_4096_50_steps20_warmup5.json", "bench_4096_50_passes_schedule.json", "bench_4096_50_f16_storage.json"]


def load(name):
    with open(os.path.join(ROOT, "profiles", ROUND, name)) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


@pytest.mark.parametrize("name", FILES)
def test_line_has_the_keys_the_driver_reads(name):
    d = load(name)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline"):
        assert k in d, k
    assert d["unit"] == "GLUPS" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "4096x4096" in d["config"]["workload"] and "50 Jacobi" in d["config"]["workload"]
    assert "model" not in d["config"]
    # value is what the timing says: cells x steps per second
    assert abs(d["value"] - 4096 * 4096 * 1e3 / d["ms_per_step"] / 1e9) <= 2e-3 * d["value"]


@pytest.mark.parametrize("name", FILES)
def test_roofline_is_a_fraction_of_a_physical_peak(name):
    r = load(name)["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 2e-3 * r["achieved"]   # bytes per launch / launch time


def test_headline_line_measures_its_traffic_and_times_the_reference_in_the_same_run():
    d = load("bench_4096_50.json")
    r = d["roofline"]
    assert "PMC" in r["traffic_source"] and r["kernel"].startswith("k_jacobi_tb")
    assert 0.9 * 12 * 4096 * 4096 <= r["traffic"] <= 1.3 * 12 * 4096 * 4096          # ten iterations for little more than the compulsory 12 B / texel
    assert d["dtype"] == "f32"
    s = d["step_hbm"]
    assert 0 < s["frac"] <= 1.0 and abs(s["frac"] - s["GBps"] / 8000.0) <= 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["unit"] == "GLUPS" and "SwiftShader" in c["renderer"] and c["sample"]
    assert 0.5 < c["steps_per_sec"] < 10 and d["steps_per_sec"] / c["steps_per_sec"] > 100
    assert d["speedup_vs_pass_structure"]["x_hbm_peak"] > 1.0      # the algorithmic-bytes figure lives under its own key, not in `roofline`


def test_the_headline_line_reproduces_from_the_committed_counter_files():
    """`roofline.traffic`, `roofline.valu` and the launch time of the committed headline line follow from the raw files committed beside it:
    the FETCH_SIZE / WRITE_SIZE passes (x 2 KiB and x 1 KiB per count), the SQ counter pass, and the kernel-trace statistics of the same command"""
    import csv
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    d = load("bench_4096_50.json")
    r = d["roofline"]
    k = r["kernel"]
    P = os.path.join(ROOT, "profiles", ROUND)
    fetch, write = pmc_traffic.per_kernel(os.path.join(P, "pmc_fetch_fused.csv")), pmc_traffic.per_kernel(os.path.join(P, "pmc_write_fused.csv"))
    per = r.get("blocks_per_dispatch", 1)   # the chained Jacobi launch: ONE dispatch of `per` blocks of iterations; the line's figures are per block
    assert abs((fetch[k][0] * 2048 + write[k][0] * 1024) / per - r["traffic"]) <= 1.0
    step = sum((fetch[n][0] * 2048 + write[n][0] * 1024) * fetch[n][1] / 4.0 for n in fetch if n in write and n.startswith("k_") and not n.startswith(("k_fill", "k_splat", "k_dye_")))   # start-up kernels, as bench.py collect_traffic
    assert abs(step - d["step_hbm"]["bytes_per_step"]) <= 8.0          # four steps under the profiler
    sq = pmc_traffic.per_kernel_counters(os.path.join(P, "pmc_sq_valu_fused.csv"))[k]
    v = r["valu"]
    assert int(sq["SQ_INSTS_VALU"][0]) == v["insts_per_launch"] and int(sq["SQ_ACTIVE_INST_VALU"][0]) == v["active_quad_cycles_per_launch"]
    simd_cycles = 4.0 * sq["SQ_ACTIVE_INST_VALU"][0] / 1024
    assert abs(simd_cycles / (sq["GRBM_GUI_ACTIVE"][0] / 8) - v["busy_frac"]) <= 2e-4
    busy = v.get("busy_frac_issue_cost", v["busy_frac"])   # lines since visit 39 weigh the count with the sweep's issue cost (bench.py JACOBI_ISSUE_COST)
    if "busy_frac_issue_cost" in v:
        assert abs(min(v["busy_frac"] * v["issue_cost_factor_model"], 1.0) - busy) <= 1e-4
    assert r["bound"] == "hbm" and (r.get("co_bound") == "valu") == (busy > r["frac_of_attainable"])
    # rocprofv3 --kernel-trace --stats of the same command: the kernel's average launch agrees with the HIP-event figure of the line (within 5 %)
    with open(os.path.join(P, "kernel_stats_fused_4096_50.csv")) as f:
        rows = [row for row in csv.DictReader(f) if k.split("<")[0] + "<" in row["Name"]]
    assert rows and abs(float(rows[0]["AverageNs"]) * 1e-6 / per - r["avg_launch_ms"]) <= 0.05 * r["avg_launch_ms"]


def test_the_documents_quote_the_drivers_latest_record():
    """VERDICT r05, housekeeping: README and RESULTS must quote the DRIVER's newest valid record (BENCH_rNN.json at the repo root: written by
    the driver at the end of a round, after the builder's last commit of that round), not only the builder's own runs under profiles/"""
    import re
    import subprocess
    # only records that are part of the history: the driver writes this round's record AFTER the round's last commit (it is untracked until
    # the next round's first commit), and no document can quote a number that does not exist yet
    try:
        tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "BENCH_r*.json"], capture_output=True, text=True, timeout=30)
    except (OSError, subprocess.SubprocessError):
        pytest.skip("no git here")
    if tracked.returncode != 0:
        pytest.skip("not a git checkout")
    recs = []
    for f in [os.path.join(ROOT, n) for n in tracked.stdout.split()]:
        with open(f) as fh:
            d = json.load(fh)
        p = d.get("parsed") or {}
        if d.get("rc") == 0 and p.get("ms_per_step") and p.get("roofline"):
            recs.append((int(re.search(r"BENCH_r(\d+)", f).group(1)), os.path.basename(f), p))
    if not recs:
        pytest.skip("no valid driver record in this checkout")
    _, name, p = max(recs)
    for doc in ("README.md", os.path.join("docs", "RESULTS.md")):
        with open(os.path.join(ROOT, doc)) as fh:
            text = fh.read()
        assert name in text, "%s does not mention %s" % (doc, name)
        assert ("%.4f" % p["ms_per_step"]) in text, "%s does not quote %s's ms_per_step %.4f" % (doc, name, p["ms_per_step"])
        assert ("%.2f" % p["value"]) in text, "%s does not quote %s's value %.2f" % (doc, name, p["value"])
