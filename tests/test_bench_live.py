"""bench.py run for real on the GPU (a small grid so that it takes seconds): ONE JSON line, the keys the driver reads, a roofline that is a
fraction of a physical peak and follows from the numbers beside it, traffic measured in the run (or a stated reason), the in-run parity
check, and the extras budget honoured.  This is the test that fails when bench.py regresses; tests/test_bench_docs_lint.py only lints
the lines committed under profiles/."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-800:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_is_live_and_consistent():
    d = run_bench("--steps", "5", "--warmup", "2", "--size", "1024", "--cpu-budget", "0")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "parity_in_run"):
        assert k in d, k
    assert d["unit"] == "GLUPS" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert "1024x1024" in d["config"]["workload"] and "50 Jacobi" in d["config"]["workload"] and "configs[1]" in d["config"]["workload"]
    assert "model" not in d["config"]
    assert abs(d["value"] - 1024 * 1024 * 1e3 / d["ms_per_step"] / 1e9) <= 2e-3 * d["value"]    # value is what the timing says
    p = d["parity_in_run"]
    assert p["ok"] and "bitwise" in p["fused_vs_passes_1024"] and "MISMATCH" not in json.dumps(p)
    r = d["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["attainable"] == 6290.0 and r["kernel"].startswith("k_jacobi_tb")
    assert 0 < r["frac"] <= 1.0 and 0 < r["frac_of_attainable"] <= 1.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert abs(r["achieved"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 2e-3 * r["achieved"]   # bytes per launch / launch time
    assert "PMC" in r["traffic_source"] or "unavailable" in r["traffic_source"]       # measured in this run, or the reason why not
    if "PMC" in r["traffic_source"]:
        assert 0.5 * 12 * 1024 * 1024 <= r["traffic"] <= 3.0 * 12 * 1024 * 1024
        s = d["step_hbm"]
        assert 0 < s["frac"] <= 1.0 and abs(s["frac"] - s["GBps"] / 8000.0) <= 1e-3
    v = r.get("valu")
    assert v is not None and ("why" in v or 0 < v["busy_frac"] <= 1.0)
    assert r["bound"] == "hbm"
    if v.get("busy_frac"):
        assert abs(min(v["busy_frac"] * v["issue_cost_factor_model"], 1.0) - v["busy_frac_issue_cost"]) <= 1e-4
        assert r["bound"] == "hbm" and (r.get("co_bound") == "valu") == (v["busy_frac_issue_cost"] > r["frac_of_attainable"])
    assert r["mem_phase_ms"] > 0 and r["valu_phase_ms"] >= 0
    assert d["steady_ms_per_step"] > 0 and d["speedup_vs_pass_structure"]["x_hbm_peak"] > 0
    # round 4: what the line says about the timed window and about what ran in it
    t = d["timed_window_regime"]
    assert len(t["ms_per_timed_step"]) == 5 and abs(sum(t["ms_per_timed_step"]) - t["sum_ms"]) < 1e-2 and t["sum_ms"] <= 5 * d["ms_per_step"] * 1.05
    # round 5: `value` is the literal W + K window (no load in front of the warm-up by default); the window behind 40 ms of load is an extra
    assert d["config"]["knobs"] == {} and "settle" not in d["config"] and "cold_start" not in d and d["effective_warmup_steps"] == 2
    assert d["config"]["build_flavor"] == "product"
    w = d["preloaded_window"]
    assert w["ms_per_step"] > 0 and w["effective_warmup_steps"] > 2
    assert all(f["equal"] for f in p["fields_1024"].values()) and len(p["fields_1024"]) == 5
    k = d["config"]["kernels"]
    assert k["chained_steps"] == 5 and k["gradsub_folded"] is True and k["curl_field_stored_by_steps"] == 2 and k["jacobi_launches_per_step"] == 5
    # (at 1024^2 the fields sit in the caches: the counters may see FEWER bytes than a launch must move, so no order between the two fractions)
    assert 0 < r["frac_compulsory"] <= 1.0 and r["compulsory_bytes_per_launch"] == 12 * 1024 * 1024
    assert abs(r["frac_compulsory"] - r["compulsory_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0) <= 2e-3
    if "step_hbm" in d:
        assert 0 < d["step_hbm"]["frac_compulsory"] <= 1.0


@pytest.mark.gpu
def test_the_drivers_command_at_the_headline_size():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` — the driver's exact command, BASELINE configs[2] (4096^2 / 50): the run that was an
    error record in round 4 (in-run parity check, packed dye: tests/test_device_view.py) must print a whole line"""
    d = run_bench("--gpus", "1", "--steps", "20", "--warmup", "5")
    assert d["value"] > 0 and d["steps"] == 20 and d["warmup"] == 5 and d["effective_warmup_steps"] == 5 and "error" not in d
    assert "configs[2]" in d["config"]["workload"] and d["config"]["kernels"]["dye_packed_rgb"] is True
    assert abs(d["value"] - 4096 * 4096 * 1e3 / d["ms_per_step"] / 1e9) <= 2e-3 * d["value"]
    p = d["parity_in_run"]
    assert p["ok"] and "bitwise" in p["fused_vs_passes_4096"] and "bitwise" in p["hip_vs_oracle_256"]
    assert all(f["equal"] for f in p["fields_4096"].values()) and len(p["fields_4096"]) == 5
    r = d["roofline"]
    assert r["kernel"].startswith("k_jacobi_tb") and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["unit"] == "GLUPS"
    assert d["preloaded_window"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_a_settled_headline_says_so():
    """--settle-ms > 0 puts the load in front of the headline itself: the line then carries config.settle, the steps that really ran in
    front of the window, and the cold window beside it"""
    d = run_bench("--steps", "5", "--warmup", "2", "--size", "1024", "--cpu-budget", "0", "--no-traffic", "--settle-ms", "20")
    assert d["config"]["settle"]["steps"] >= 1 and d["effective_warmup_steps"] == 2 + d["config"]["settle"]["steps"] + 6
    assert d["cold_start"]["ms_per_step"] > 0 and len(d["cold_start"]["ms_per_timed_step"]) == 5 and "preloaded_window" not in d


@pytest.mark.gpu
def test_bench_extras_budget_is_honoured():
    """with no budget left for the extras the line still comes, at once, and says what it skipped"""
    d = run_bench("--steps", "3", "--warmup", "1", "--size", "512", "--cpu-budget", "5", "--extras-budget", "1", "--no-parity")
    assert d["value"] > 0 and "roofline" in d
    assert "model" in d["roofline"]["traffic_source"] and "budget" in d["roofline"]["traffic_source"]
    assert d["cpu_baseline"]["value"] is None and "budget" in d["cpu_baseline"]["sample"]
    assert d["skipped"]["steady_ms_per_step"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,tiles_x,canvas", [(4, 1, (256, 1024)), (4, 2, (512, 512))])
def test_the_decomposition_checker_of_the_multi_gpu_line(world, tiles_x, canvas):
    """bench.py N > 1 checks in the run that a fresh stripe / tile set over RCCL leaves the single domain's bits on every rank's own rows
    (decomposition_in_run).  Its comparison — owned rows x columns of each context against the global field, on the device — runs here on an
    in-process set: equal as the library's decomposition is, and a context that was tampered with is caught with its rows named."""
    import numpy as np
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    sys.path.insert(0, ROOT)
    import bench
    res = min(canvas)
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": res, "PRESSURE_ITERATIONS": 20}
    g = StripeGroup(world, canvas=canvas, config=cfg, halo=24, random=fluid_hip.mulberry32(1234), tiles_x=tiles_x)
    try:
        with fluid_hip.FluidSim(canvas=canvas, config=cfg, random=fluid_hip.mulberry32(1234)) as one:
            for sim in (g, one):
                sim.multipleSplats(6)
                sim.step(bench.DT, 3)
            f = bench.compare_with_single_domain(one, g.engines, 0)
            assert set(f) == {"velocity", "pressure", "divergence", "curl", "dye"} and all(v["equal"] for v in f.values()), f
            e = g.engines[world - 1]
            p = e.read("pressure")
            p[3, 5] += 1.0
            e.write("pressure", p)
            f = bench.compare_with_single_domain(one, g.engines, 0)
            fi = e.info("pressure")
            assert not f["pressure"]["equal"] and f["pressure"]["n_diff"] == 1 and f["pressure"]["rows"] == [fi.row0, fi.row0 + fi.rows]
            assert f["pressure"]["first_diff_at"][:2] == [3, 5] and all(v["equal"] for k, v in f.items() if k != "pressure")
    finally:
        g.close()

