"""DOCS LINT, not a test of bench.py (that is tests/test_bench_live.py, which runs it): the bench lines COMMITTED under profiles/ — the
numbers DESIGN.md quotes — are well-formed lines of the contract (profiles/r03/bench_4096_50.json is what `python bench.py` printed there): the keys the driver reads, a roofline that is a fraction of a physical peak (<= 1, = achieved / peak,
achieved = measured HBM bytes per launch / measured launch time), the CPU baseline of the same run and what kind it is."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["bench_4096_50.json", "bench_4096_50_steps20_warmup5.json", "bench_4096_50_passes_schedule.json", "bench_4096_50_f16_storage.json"]


def load(name):
    with open(os.path.join(ROOT, "profiles", "r03", name)) as f:
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
