#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X stable-fluids hot path.

Metric (BASELINE.json): sim steps/s and cell-updates/s (GLUPS = W*H*steps/s / 1e9) at 4096^2,
50 Jacobi iterations per step, fp32, dye resolution = sim resolution.

A "step" is one reference step(dt) (script.js:1231-1294) over the whole grid.  Inputs are resident in
HBM before the timed region (20 seeded splats applied on the device); the timed region is K steps
enqueued back to back, bracketed by barrier + device sync on both sides, max over ranks.

  N = 1 : one whole-domain context, 4096 x 4096.
  N > 1 : weak scaling — N row stripes of 4096 x 4096 each (global grid 4096 x 4096*N), one process
          per GPU, ghost rows exchanged with ncclSend/ncclRecv (RCCL over xGMI) issued by libfluid_hip.so
          itself; torch.distributed carries the ncclUniqueId, the barrier and the max-over-ranks time.
          No collective on the data path other than neighbour exchange.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "webgl-fluid-simulation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
DT = 0.016666           # the reference's dt clamp, script.js:1191


def algorithmic_bytes_per_cell(iters: int) -> int:
    # curl 12 + vorticity 20 + divergence 12 + clear 8 + Jacobi 12/iter + gradsub 20
    # + advect velocity 16 + advect dye 40 (fp32, dye res = sim res, RGBA dye)
    return 128 + 12 * iters


def cpu_baseline_reference(size: int, iters: int, timeout_s: float):
    """The reference ITSELF — the unmodified script.js step() (script.js:1231-1294) under Chromium + SwiftShader (software WebGL,
    kaleido package) — timed on this host's cores on the same workload: 3 warm-up + 5 timed whole steps with a readPixels sync per
    step (oracle/live/time_reference.py).  The page script is the staged byte-for-byte copy oracle/_ref/ (oracle/stage_reference.sh;
    /root/reference in the build container).  Runs in its own process group under a timeout; returns (dict | None, reason)."""
    import signal
    import subprocess
    script = os.path.join(ROOT, "oracle", "live", "time_reference.py")
    if size > 8192:
        return None, "the reference cannot run above 8192^2 (SwiftShader MAX_TEXTURE_SIZE)"
    try:
        import kaleido  # noqa: F401
    except Exception as ex:
        return None, "kaleido (Chromium + SwiftShader) is not importable here: %s" % ex
    have = [d for d in ("/root/reference", os.path.join(ROOT, "oracle", "_ref")) if os.path.exists(os.path.join(d, "script.js"))]
    if not have:
        return None, "no reference page script on this box (neither /root/reference nor the staged oracle/_ref/)"
    p = subprocess.Popen([sys.executable, script, "--size", str(size), "--iters", str(iters), "--warm", "3", "--timed", "5", "--json"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        so, se = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)   # exactly the process group started above (python + the browser it spawned)
        except OSError:
            pass
        p.wait()
        return None, "the live reference did not finish within %.0f s" % timeout_s
    lines = [l for l in so.decode(errors="replace").splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return None, "the live reference failed here: " + (se.decode(errors="replace").strip().splitlines() or ["no output"])[-1][:200]
    r = json.loads(lines[-1])
    return {"value": r["GLUPS"], "unit": "GLUPS", "steps_per_sec": r["steps_per_sec"], "ms_per_step": r["ms_per_step"],
            "cores": r["nproc"], "kind": "reference", "cpu_model": r["cpu_model"],
            "renderer": "%s / %s" % (r["gl"].get("renderer"), r["gl"].get("version")), "user_agent": r["gl"].get("userAgent"),
            "swiftshader_threads": r["gl"].get("cores"),
            "sample": "%d whole step(s) of the same %dx%d / %d-iteration workload after %d warm-up steps: the unmodified reference "
                      "script.js step() under headless Chromium + SwiftShader (software WebGL2), readPixels sync per step; "
                      "`cores` = host cores available, SwiftShader keeps only a few of them busy"
                      % (r["timed_steps"], r["sim"][0], r["sim"][1], iters, r["warmup_steps"])}, None


def cpu_baseline_port(size: int, iters: int, budget_s: float):
    """The CPU oracle (a port of the reference's algorithm, OpenMP over rows) timed on this host's
    cores on a bounded sample of the same workload: same grid, same splats, whole steps."""
    from oracle import oracle as O
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    ref = O.RefSim(canvas=(size, size), config=cfg, seed=1234)
    ref.multiple_splats(20)
    t0 = time.perf_counter()
    ref.step(DT, 1)  # first step also pays first-touch; kept if it is the only one
    first = time.perf_counter() - t0
    steps, spent = 0, 0.0
    while spent < budget_s and steps < 50:
        t0 = time.perf_counter()
        ref.step(DT, 1)
        spent += time.perf_counter() - t0
        steps += 1
        if spent + spent / steps > budget_s:
            break
    per = spent / steps if steps else first
    return {"value": round(size * size / per / 1e9, 6), "unit": "GLUPS", "steps_per_sec": round(1.0 / per, 4),
            "cores": O.num_threads(), "kind": "port",
            "sample": "%d whole step(s) of the same %dx%d / %d-iteration workload after 1 warm-up step, oracle/fluid_oracle.c (OpenMP)"
                      % (steps or 1, size, size, iters)}


def cpu_baseline(size: int, iters: int, budget_s: float, kind: str = "auto"):
    """`cpu_baseline` of the JSON line: the live reference when it can run on this box (kind "reference"), else the C/OpenMP port
    of its algorithm (kind "port") with the reason the reference could not run.  With the reference as the baseline the port's
    number is still reported beside it (`port`), on a short sample, for the record."""
    why = None
    if kind in ("auto", "reference"):
        ref, why = cpu_baseline_reference(size, iters, timeout_s=max(240.0, 12 * budget_s))
        if ref is not None:
            if budget_s > 0:
                port = cpu_baseline_port(size, iters, min(budget_s, 6.0))
                ref["port"] = {k: port[k] for k in ("value", "unit", "steps_per_sec", "cores", "sample")}
            return ref
        if kind == "reference":
            return {"value": None, "unit": "GLUPS", "kind": "reference", "cores": os.cpu_count(), "sample": "not measured: " + why}
    out = cpu_baseline_port(size, iters, budget_s)
    if why:
        out["sample"] += "; the live reference could not be timed here: " + why
    return out


def collect_traffic(args, steps_under_profiler: int = 4):
    """HBM bytes per launch of every step kernel, measured IN THIS RUN: two short child runs of this very script under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE in separate passes, as the guide's HBM section
    prescribes; no other tracing domain), corrected as tools/pmc_traffic.py documents (KiB units, x2 on FETCH_SIZE for gfx950,
    WRITE_SIZE calibrated to 1.0 on k_clear in profiles/r01).  Returns {kernel: {...}} + per-step totals, or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(prefix="fluid_pmc_", dir="/tmp") as d:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", str(steps_under_profiler - 1), "--warmup", "1", "--cpu-budget", "0", "--no-profile-pass",
                   "--no-traffic", "--no-steady", "--size", str(args.size), "--iters", str(args.iters), "--schedule", args.schedule,
                   "--storage", args.storage]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s did not finish within 420 s" % ctr
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (ctr, r.returncode, (r.stderr.decode(errors="replace").strip().splitlines() or [""])[-1][:160])
            per[ctr] = pmc_traffic.per_kernel(files[0])
            keep = os.environ.get("FLUID_BENCH_KEEP_PMC")   # tools/gpu_round.sh: keep the raw counter CSVs for profiles/
            if keep:
                shutil.copy(files[0], os.path.join(keep, "pmc_%s_%s.csv" % (ctr, args.schedule)))
    kernels, step_bytes = {}, 0.0
    for k in sorted(set(per["FETCH_SIZE"]) & set(per["WRITE_SIZE"])):
        if not k.startswith("k_") or k.startswith("k_fill") or k.startswith("k_splat"):
            continue   # start-up kernels (fills, splats) are not part of a step
        rd = per["FETCH_SIZE"][k][0] * 1024.0 * 2.0
        wr = per["WRITE_SIZE"][k][0] * 1024.0 * 1.0
        n = per["FETCH_SIZE"][k][1]
        kernels[k] = {"read_bytes": int(rd), "write_bytes": int(wr), "bytes_per_launch": int(rd + wr), "launches_per_step": n / steps_under_profiler}
        step_bytes += (rd + wr) * n / steps_under_profiler
    return {"kernels": kernels, "bytes_per_step": int(step_bytes)}, None


class Watchdog:
    """N > 1: if the communicator set-up or the first exchanges hang (a wedged RCCL / xGMI link), every rank would sit in a
    collective until the driver's timeout and leave nothing to diagnose.  The watchdog prints ONE JSON line with `error` and
    the stage that did not complete, then ends the process."""

    def __init__(self, fd, seconds, base):
        import threading
        self.fd, self.seconds, self.base, self.stage = fd, seconds, base, "start"
        self.done = threading.Event()
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        if not self.done.wait(self.seconds):
            out = dict(self.base, value=None, error="watchdog: stage '%s' did not complete within %.0f s on rank %s"
                       % (self.stage, self.seconds, os.environ.get("RANK", "0")))
            os.write(self.fd, (json.dumps(out) + "\n").encode())
            os._exit(3)

    def at(self, stage):
        self.stage = stage

    def stop(self):
        self.done.set()


def main(argv=None, engine_factory=None, backend="nccl"):
    """`engine_factory` / `backend` are the hooks of tests/test_bench_multi.py: the N > 1 branch of THIS function — rendezvous,
    StripeSim set-up, barrier, max-over-ranks timing, the JSON line — runs on CPU ranks over gloo with an injected stripe engine.
    The command line cannot select either: `python bench.py` always measures libfluid_hip.so on GPUs over RCCL."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the GPU needs ~30 ms of work to reach its steady clocks (profiles/r01/bench_warmup_sensitivity.txt: with 50 warm-up
    # steps even 5 timed steps read the steady 0.539 ms/step; with 3 they read 0.595), so warm up for 50 steps and time 200
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--size", type=int, default=4096, help="grid edge per GPU (BASELINE config: 4096)")
    ap.add_argument("--iters", type=int, default=50, help="PRESSURE_ITERATIONS (BASELINE config: 50)")
    ap.add_argument("--schedule", default="fused", choices=["fused", "passes"])
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"], help="field storage; f32 is the headline, f16 (the reference's "
                    "half-float textures, SURVEY 8f N4) is a side measurement and says so in the JSON line")
    ap.add_argument("--halo", type=int, default=56, help="ghost rows per stripe side (N > 1); >= 54 keeps 50 Jacobi iterations in one "
                                                        "block: 2 exchanges per step (profiles/r01/stripe_overhead_one_gpu.txt)")
    ap.add_argument("--reach", type=int, default=32, help="N > 1: ghost rows refreshed in front of the advection (rows a back-trace may span); "
                    "the 4096 x 32768 grid of the 8-rank run reaches |v| = 1106 = 18.4 rows (tools/max_velocity.py), the library default is 24")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-kind", default="auto", choices=["auto", "reference", "port"], help="cpu_baseline: the live reference under "
                    "SwiftShader when it can run here (auto / reference), or the C/OpenMP port of its algorithm")
    ap.add_argument("--no-profile-pass", action="store_true", help="skip the HIP-event instrumented pass")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure HBM bytes per launch")
    ap.add_argument("--no-steady", action="store_true", help="skip the long (>= 2000 steps) steady-state timing appended to the line")
    ap.add_argument("--tiles-x", type=int, default=1, help="N > 1: 2-D decomposition, N // tiles_x row stripes x tiles_x column tiles "
                                                           "(global grid size*tiles_x x size*N/tiles_x); default 1 = row stripes")
    ap.add_argument("--strong", action="store_true", help="N > 1: strong scaling — the global grid stays --size x --size and is cut into N "
                                                          "stripes / tiles (BASELINE configs[3]: --size 8192 --tiles-x 2 on 4 GPUs; configs[4]: "
                                                          "--size 16384 --iters 200 on 8 GPUs); default: weak scaling, --size x --size per GPU")
    ap.add_argument("--hosted", action="store_true", help="N > 1: drive the passes from Python with torch.distributed send/recv "
                                                          "instead of the native plan + RCCL inside libfluid_hip.so")
    ap.add_argument("--comm-timeout", type=float, default=120.0, help="N > 1: seconds the communicator set-up and the warm-up steps may take "
                                                                     "before the watchdog reports which stage hung")
    args = ap.parse_args(argv)
    on_cpu = engine_factory is not None   # only tests/test_bench_multi.py: the launcher path on CPU ranks

    # stdout must carry exactly ONE JSON line: RCCL / HIP libraries print banners to fd 1 from C, so everything
    # else is routed to stderr and the JSON is written to the saved descriptor at the end
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # before the HIP / HSA runtime initialises: the host driver only supports dmabuf IPC (RCCL across processes needs it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import fluid_hip

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N, size, iters = world, args.size, args.iters
    base = {"metric": "cell-updates/sec (GLUPS) at %d^2 per GPU, %d Jacobi iters/step" % (size, iters), "value": None, "unit": "GLUPS",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True}

    def fail(msg, code=2):
        print("bench.py: " + msg, file=sys.stderr)
        if rank == 0:
            os.write(real_stdout, (json.dumps(dict(base, error=msg)) + "\n").encode())
        sys.exit(code)

    if world != args.gpus:
        fail("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE is %d)" % (args.gpus, args.gpus, world))
    if not on_cpu:
        if not torch.cuda.is_available():
            fail("no GPU visible; the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
    dev_sync = (lambda: None) if on_cpu else torch.cuda.synchronize
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    dog = None

    if N == 1:
        sim = fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=local_rank, schedule=args.schedule,
                                 random=fluid_hip.mulberry32(1234), storage=args.storage)
        sim.multipleSplats(20)
        barrier = lambda: None  # noqa: E731
        grid_w, grid_h = size, size
    else:
        import torch.distributed as dist
        from fluid_hip.stripes import StripeSim
        dog = Watchdog(real_stdout, args.comm_timeout, base)
        dog.at("torch.distributed rendezvous (init_process_group, backend %s)" % backend)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {} if on_cpu else {"device_id": torch.device("cuda", local_rank)}
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        # global grid: `size` columns x `size * N` rows -> canvas of the same aspect, SIM_RESOLUTION = short side
        # (--tiles-x T: size * T columns x size * N / T rows, every rank still owns size x size texels)
        tx = max(1, args.tiles_x)
        gw, gh = (size, size) if args.strong else (size * tx, size * N // tx)
        cfg = dict(cfg, SIM_RESOLUTION=min(gw, gh), DYE_RESOLUTION=min(gw, gh))
        dog.at("communicator set-up (ncclGetUniqueId on rank 0, broadcast, ncclCommInitRank inside libfluid_hip.so)")
        try:
            kw = dict(engine_factory=engine_factory) if on_cpu else dict(native=not args.hosted, tiles_x=tx, storage=args.storage,
                                                                        reach=min(args.reach, args.halo))
            sim = StripeSim(canvas=(gw, gh), config=cfg, halo=args.halo, schedule=args.schedule, random=fluid_hip.mulberry32(1234),
                            device=local_rank, **kw)
        except Exception as ex:   # no silent switch to another driver: say what failed, on every rank, and stop
            dog.stop()
            fail("stripe driver set-up failed on rank %d: %s" % (rank, ex), code=4)
        sim.multipleSplats(20)
        barrier = dist.barrier
        grid_w, grid_h = gw, gh

    def run(k):
        sim.step(DT, k)   # N > 1, native driver: the plan and its RCCL exchanges run inside libfluid_hip.so

    def sync():
        sim.sync()
        dev_sync()

    if dog:
        dog.at("warm-up: %d steps (the first ghost-row exchanges over RCCL / xGMI)" % args.warmup)
    try:
        run(args.warmup)
        sync(); barrier(); sync()
        if dog:
            dog.stop()
        t0 = time.perf_counter()
        run(args.steps)
        sync(); barrier(); sync()
        elapsed = time.perf_counter() - t0
    except fluid_hip.FluidError as ex:
        if dog:
            dog.stop()
        fail("step failed on rank %d: %s" % (rank, ex), code=5)
    if N > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        sim.check_halo()

    steps_per_s = args.steps / elapsed
    glups = grid_w * grid_h * steps_per_s / 1e9
    alg_step_bytes = algorithmic_bytes_per_cell(iters) * grid_w * grid_h * (0.5 if args.storage == "f16" else 1.0)
    out = dict(base)
    out.update({
        "value": round(glups, 4),
        "steps_per_sec": round(steps_per_s, 3),
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "scaling": "strong" if (args.strong and N > 1) else "weak", "vs_baseline": None,
        "dtype": "f32" if args.storage == "f32" else "f32 arithmetic on f16-stored fields (side measurement, not the headline)",
        "data": "synthetic",
        "config": {"workload": "configs[2]: %dx%d sim = dye grid%s, %d Jacobi iters/step, dt=%.6f, 20 splats mulberry32(1234), defaults otherwise"
                               % (grid_w, grid_h, "" if N == 1 else " (%d ranks of %dx%d, halo %d)" % (N, grid_w // max(1, args.tiles_x), grid_h * max(1, args.tiles_x) // N, args.halo), iters, DT),
                   "schedule": args.schedule, "storage": args.storage,
                   "parallelism": "single" if N == 1 else ("stripes%d" % N if args.tiles_x <= 1 else "tiles%dx%d" % (N // args.tiles_x, args.tiles_x))},
        # what the reference's pass structure would have to move for this many steps per second (SURVEY 8d's byte model): with
        # temporal blocking this is a speed-up figure, NOT a fraction of the HBM roofline — the bounded fractions are below
        "speedup_vs_pass_structure": {"algorithmic_GBps": round(alg_step_bytes * steps_per_s / 1e9, 1),
                                      "x_hbm_peak": round(alg_step_bytes * steps_per_s / 1e9 / (HBM_PEAK_GBPS * N), 4)},
    })
    if on_cpu:
        out["config"]["engine"] = "injected stripe engine on CPU ranks over %s (launcher-path test, not a measurement)" % backend

    # ---- the dominant kernel (the Jacobi loop): launch time from HIP events on the solver's own stream, HBM bytes from PMC passes ----
    if rank == 0 and N == 1 and not args.no_profile_pass:
        sim.set_timing(True)
        sim.step(DT, min(args.steps, 20))
        sim.sync()
        tm = sim.timings()
        sim.set_timing(False)
        # launches of the loop that are ONLY Jacobi (the last launch of a step also carries the gradient subtract under the fused
        # schedule: k_jacobi_tb_gs, timed under gradsub_ms)
        launches = max(tm["jacobi_launches"] - tm.get("folded_launches", 0), 1)
        avg_ms = tm["jacobi_ms"] / launches
        half = 0.5 if args.storage == "f16" else 1.0
        alg_launch = 12.0 * iters * size * size * tm["steps"] / max(tm["jacobi_launches"], 1) * half  # 12 B/cell/iteration, SURVEY.md 8(d)
        # the kernel that runs the loop: the temporally blocked register tile, or one launch per iteration under --schedule passes
        if args.schedule == "fused":
            cands = ["k_jacobi_tb_h<", "k_jacobi_tb<"] if args.storage == "f16" else ["k_jacobi_tb<"]
        else:
            cands = ["k_h_jacobi", "k_jacobi"] if args.storage == "f16" else ["k_jacobi"]
        kname = cands[0]
        traffic, why = (None, "--no-traffic") if args.no_traffic else collect_traffic(args)
        entry = None
        if traffic:
            for c in cands:
                hit = [(k, v) for k, v in traffic["kernels"].items() if k.startswith(c)]
                if hit:
                    kname, entry = hit[0]
                    break
            if entry is None:
                why = "no %s dispatch in the counter pass" % cands[0]
        if entry:
            bytes_launch, source = entry["bytes_per_launch"], "PMC FETCH_SIZE x2 + WRITE_SIZE, this run (rocprofv3 --pmc, separate passes)"
        else:   # the least a launch must move: pressure in, divergence in, pressure out (no apron re-reads counted)
            bytes_launch, source = int(12.0 * size * size * half), "model: compulsory 12 B/texel per launch (PMC pass unavailable: %s)" % why
        achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {
            "kernel": kname, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": bytes_launch, "traffic_source": source,
            "avg_launch_ms": round(avg_ms, 5), "launches_per_step": launches / max(tm["steps"], 1),
            "iterations_per_launch": iters * tm["steps"] / max(tm["jacobi_launches"], 1),
            "algorithmic_bytes_per_launch": int(alg_launch),
            "algorithmic_GBps": round(alg_launch / (avg_ms * 1e-3) / 1e9, 1),
            "note": "achieved = HBM bytes one launch really moves / its measured duration (bounded by the peak); "
                    "algorithmic_* = the reference's 12 B/cell/iteration for the iterations this launch performs (a speed-up over the "
                    "pass structure, may exceed the peak)",
        }
        if traffic:
            out["step_hbm"] = {"bytes_per_step": traffic["bytes_per_step"], "GBps": round(traffic["bytes_per_step"] * steps_per_s / 1e9, 1),
                               "frac": round(traffic["bytes_per_step"] * steps_per_s / 1e9 / HBM_PEAK_GBPS, 4),
                               "kernels": {k: {"bytes_per_launch": v["bytes_per_launch"], "launches_per_step": v["launches_per_step"]}
                                           for k, v in traffic["kernels"].items()}}
        per_step = {k: round(v / max(tm["steps"], 1), 4) for k, v in tm.items() if k.endswith("_ms")}
        out["pass_ms_per_step"] = per_step

    # ---- the same loop well inside steady clocks: the contract's K steps may be as few as 20 (11 ms), inside the clock ramp ----
    if rank == 0 and N == 1 and not args.no_steady:
        n_long = max(2000, args.steps)
        sync()
        t0 = time.perf_counter()
        run(n_long)
        sync()
        out["steady_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / n_long, 4)
        out["steady_steps"] = n_long

    if rank == 0 and N == 1 and args.cpu_budget > 0:
        out["cpu_baseline"] = cpu_baseline(size, iters, args.cpu_budget, args.cpu_kind)

    if N > 1:
        out["config"]["exchanges_per_step"] = sim.exchanges / max(args.steps + args.warmup, 1)
        out["config"]["driver"] = "native plan + ncclSend/ncclRecv inside libfluid_hip.so" if sim.native else "hosted: torch.distributed batch_isend_irecv"
        if sim.native:
            out["config"]["advect_exchange_rows"] = list(sim.engine.advect_exchange_rows())
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
