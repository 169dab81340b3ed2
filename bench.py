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

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak (spec), /opt/skills/guides/MI355X_MICROARCH.md
HBM_ATTAINABLE_GBPS = 6290.0  # the same guide: measured float4 copy, 79 % of the peak
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
    timed = [float(x) for x in r.get("ms", [])][r.get("warmup_steps", 0):]
    spread = {"ms_per_step_min": round(min(timed), 1), "ms_per_step_max": round(max(timed), 1)} if timed else {}
    return {"value": r["GLUPS"], "unit": "GLUPS", "steps_per_sec": r["steps_per_sec"], "ms_per_step": r["ms_per_step"], **spread,
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


def cpu_baseline(size: int, iters: int, budget_s: float, kind: str = "auto", max_wait: float = 1e9):
    """`cpu_baseline` of the JSON line: the live reference when it can run on this box (kind "reference"), else the C/OpenMP port
    of its algorithm (kind "port") with the reason the reference could not run.  With the reference as the baseline the port's
    number is still reported beside it (`port`), on a short sample, for the record."""
    why = None
    if kind in ("auto", "reference"):
        ref, why = cpu_baseline_reference(size, iters, timeout_s=min(max(240.0, 12 * budget_s), max(30.0, max_wait - 10.0)))
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


def link_arg(spec):
    """--link-model: "US,GBPS" -> that model; "default" -> the library's constants; None / "calibrate" -> measured at start-up"""
    if spec in (None, "calibrate"):
        return "calibrate"
    if spec == "default":
        return None
    return tuple(float(x) for x in spec.split(","))


def fluid_knobs():
    """every FLUID_* variable in the environment: the library's A/B knobs (tile shapes, folds, chains, another build of the library, an
    RCCL stand-in) change what is timed, so the line lists them (`config.knobs`; empty = the shipped defaults)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("FLUID_") and k not in ("FLUID_BENCH_KEEP_PMC",)}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves — the driver's own command line,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>` —
    and pass rank 0's one JSON line through.  `script` is what the ranks run (this file; tests/test_bench_multi.py passes the entry that
    injects CPU ranks).  Returns the exit code; if the ranks die without a line, prints an error line that says how many GPUs were asked for."""
    import subprocess
    script = script or os.path.abspath(__file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=timeout)
        rc, so = r.returncode, r.stdout.decode(errors="replace")
    except subprocess.TimeoutExpired as ex:
        rc, so = 124, (ex.stdout or b"").decode(errors="replace")
    lines = [l for l in so.splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
    else:
        sys.stdout.write(json.dumps({"metric": "cell-updates/sec (GLUPS)", "value": None, "unit": "GLUPS", "n_gpus": n, "higher_is_better": True,
                                     "error": "the %d ranks started by bench.py itself (torch.distributed.run) ended with exit code %d and no JSON line" % (n, rc)}) + "\n")
    sys.stdout.flush()
    return rc if (rc or lines) else 1


class Deadline:
    """The extras behind the timed section (PMC child runs, steady timing, the CPU baseline) share one budget (--extras-budget): the
    headline line must not wait for a slow rocprofv3 or SwiftShader; an extra that no longer fits is skipped and says so."""

    def __init__(self, seconds):
        self.t_end = time.monotonic() + seconds

    def left(self):
        return self.t_end - time.monotonic()


def pmc_pass(args, counters, deadline, steps_under_profiler=4):
    """One child run of this very script under `rocprofv3 --kernel-trace --pmc <counters>` (no other tracing domain): average counter
    value per dispatch of every kernel -> ({kernel: {counter: (avg, dispatches)}}, None) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 is not on PATH"
    limit = min(240.0, deadline.left())
    if limit < 45:
        return None, "extras budget spent before the %s pass" % "+".join(counters)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    with tempfile.TemporaryDirectory(prefix="fluid_pmc_", dir="/tmp") as d:
        cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
               os.path.abspath(__file__), "--steps", str(steps_under_profiler), "--warmup", "0", "--cpu-budget", "0", "--no-profile-pass",
               "--no-traffic", "--no-steady", "--no-parity", "--settle-ms", "0", "--size", str(args.size), "--iters", str(args.iters), "--schedule", args.schedule,
               "--storage", args.storage]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit)
        except subprocess.TimeoutExpired:
            return None, "rocprofv3 --pmc %s did not finish within %.0f s" % (" ".join(counters), limit)
        files = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
        if r.returncode != 0 or not files:
            return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), r.returncode, (r.stderr.decode(errors="replace").strip().splitlines() or [""])[-1][:160])
        out = pmc_traffic.per_kernel_counters(files[0])
        keep = os.environ.get("FLUID_BENCH_KEEP_PMC")   # tools/gpu_round.sh: keep the raw counter CSVs for profiles/
        if keep:
            shutil.copy(files[0], os.path.join(keep, "pmc_%s_%s.csv" % ("_".join(counters)[:40], args.schedule)))
        return out, None


def collect_traffic(args, deadline, steps_under_profiler: int = 0, chained: bool = False):
    """HBM bytes per launch of every step kernel, measured IN THIS RUN: two short child runs under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE and WRITE_SIZE in separate passes, as the guide's HBM section prescribes), corrected as tools/pmc_traffic.py documents
    (KiB units, x2 on FETCH_SIZE for gfx950, WRITE_SIZE calibrated to 1.0 on k_clear in profiles/r01).
    Returns ({kernels, bytes_per_step}, None) or (None, reason)."""
    if not steps_under_profiler:
        # ONE call of this many steps.  Where a call for n steps chains them (k_advect_cvd; fluid_schedule_info says so), sixteen steps put the
        # launch mix within 6 % of the timed call's (the bytes per launch are what they are either way)
        steps_under_profiler = 16 if chained else 4
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        got, why = pmc_pass(args, [ctr], deadline, steps_under_profiler)
        if got is None:
            return None, why
        per[ctr] = {k: v[ctr] for k, v in got.items() if ctr in v}
    kernels, step_bytes = {}, 0.0
    for k in sorted(set(per["FETCH_SIZE"]) & set(per["WRITE_SIZE"])):
        if not k.startswith("k_") or k.startswith("k_fill") or k.startswith("k_splat") or k.startswith("k_dye_"):
            continue   # start-up kernels (fills, splats, the one-off packing of the dye field) are not part of a step
        rd = per["FETCH_SIZE"][k][0] * 1024.0 * 2.0
        wr = per["WRITE_SIZE"][k][0] * 1024.0 * 1.0
        n = per["FETCH_SIZE"][k][1]
        kernels[k] = {"read_bytes": int(rd), "write_bytes": int(wr), "bytes_per_launch": int(rd + wr), "launches_per_step": n / steps_under_profiler}
        step_bytes += (rd + wr) * n / steps_under_profiler
    return {"kernels": kernels, "bytes_per_step": int(step_bytes)}, None


N_SIMD = 256 * 4   # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
JACOBI_ISSUE_COST = round((9 * 1.67 + 2 * 1.55) / 11, 2)   # issue slots per instruction of the Jacobi sweep, relative to v_fma_f32 (see roofline.valu)


def collect_valu(args, deadline, kernel_prefixes):
    """What the dominant kernel's VALU pipes did, from one more counter pass in this run: SQ_INSTS_VALU (wave-instructions),
    SQ_ACTIVE_INST_VALU (quad-cycles a wave spent issuing VALU, summed over waves; x4 = cycles), SQ_WAVE_CYCLES, SQ_BUSY_CU_CYCLES and
    GRBM_GUI_ACTIVE (the dispatch's duration in shader-clock cycles).  valu_busy_frac = the average SIMD's VALU-issuing cycles / the
    dispatch's cycles: the fraction of the launch during which the kernel was paying for arithmetic, whatever else it overlapped with."""
    got, why = pmc_pass(args, ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"], deadline)
    if got is None:
        return None, why
    for pre in kernel_prefixes:
        for k, v in got.items():
            if k.startswith(pre) and "SQ_ACTIVE_INST_VALU" in v:
                insts, active = v["SQ_INSTS_VALU"][0], v["SQ_ACTIVE_INST_VALU"][0]
                gui = v.get("GRBM_GUI_ACTIVE", (0, 0))[0]
                out = {"kernel": k, "insts_per_launch": int(insts), "active_quad_cycles_per_launch": int(active),
                       "wave_cycles_per_launch": int(v.get("SQ_WAVE_CYCLES", (0, 0))[0]), "gui_active_cycles_per_launch": int(gui),
                       "cycles_per_inst": round(4.0 * active / max(insts, 1), 3),
                       "simd_valu_cycles_per_launch": int(4.0 * active / N_SIMD)}
                if gui > 0:
                    out["busy_frac"] = round(4.0 * active / N_SIMD / gui, 4)
                return out, None
    return None, "no %s dispatch in the SQ counter pass" % kernel_prefixes[0]


class Watchdog:
    """N > 1: if the communicator set-up or the first exchanges hang (a wedged RCCL / xGMI link), every rank would sit in a
    collective until the driver's timeout and leave nothing to diagnose.  The watchdog prints ONE JSON line with `error` and
    the stage that did not complete, then ends the process."""

    def __init__(self, fd, seconds, base):
        import threading
        self.fd, self.seconds, self.base, self.stage = fd, seconds, base, "start"
        self.result = None   # the finished headline measurement: what hangs BEHIND it (an extra configuration) must not take it away
        self.deadline = time.time() + seconds   # every stage gets `seconds` of its own (at() moves it)
        self.done = threading.Event()
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        while not self.done.wait(0.5):
            if time.time() < self.deadline:
                continue
            msg = "watchdog: stage '%s' did not complete within %.0f s on rank %s" % (self.stage, self.seconds, os.environ.get("RANK", "0"))
            if self.result is not None:   # the weak-scaling number stands; the line says what did not finish behind it
                if os.environ.get("RANK", "0") == "0":
                    out = dict(self.result)
                    out["extra_configs"] = list(out.get("extra_configs", [])) + [{"error": msg}]
                    out.setdefault("parity_in_run", "not run: " + msg)
                    os.write(self.fd, (json.dumps(out) + "\n").encode())
                    os._exit(0)
                time.sleep(3.0)   # rank 0 prints the line first; a launcher that sees a failed worker ends the others
                os._exit(3)
            out = dict(self.base, value=None, error=msg)
            os.write(self.fd, (json.dumps(out) + "\n").encode())
            os._exit(3)

    def at(self, stage):
        self.stage = stage
        self.deadline = time.time() + self.seconds

    def stop(self):
        self.done.set()


def parity_in_run(fluid_hip, size, iters, device, storage, with_oracle=True, steps=10):
    """Parity checked in the same run as the number, after the timed steps:
      (a) HIP == the CPU oracle bit for bit on a small case (256^2 sim / 512^2 dye, 20 iterations, 6 seeded splats, 2 steps, both
          schedules) — the oracle is the checker here, never the thing timed;
      (b) at the benchmark's own size, on this rank's GPU: the fused schedule (what is timed) == the one-kernel-per-reference-pass
          schedule, every field, compared on the device after `steps` steps from the same seeded splats.
    It runs BEHIND the timed section: in front of it, it made the driver's `--steps 20 --warmup 5` read 3-4 % slower (three contexts of
    1 GB alive and 30 ms of load straight before the warm-up; profiles/r03/driver_flags_preroll.txt) — the gap between that figure and the
    steady one is not a clock ramp that prior load removes.  A mismatch is an error of the run, not a footnote."""
    import numpy as np
    import torch
    out = {}
    t0 = time.perf_counter()
    if with_oracle:
        try:
            from oracle import oracle as O
            cfg = {"SIM_RESOLUTION": 256, "DYE_RESOLUTION": 512, "PRESSURE_ITERATIONS": 20}
            ref = O.RefSim(canvas=(512, 512), config=cfg, seed=4321, storage=storage) if storage == "f16" else O.RefSim(canvas=(512, 512), config=cfg, seed=4321)
            ref.multiple_splats(6)
            ref.step(DT, 2)
            want = ref.fields()
            ok = True
            for schedule in ("passes", "fused"):
                with fluid_hip.FluidSim(canvas=(512, 512), config=cfg, device=device, schedule=schedule, random=fluid_hip.mulberry32(4321),
                                        storage=storage) as sim:
                    sim.multipleSplats(6)
                    sim.step(DT, 2)
                    got = sim.fields()
                ok = ok and all(np.array_equal(got[k], w) for k, w in want.items())
            out["hip_vs_oracle_256"] = "bitwise equal, both schedules" if ok else "MISMATCH"
        except Exception as ex:   # the oracle library may be absent on a box: say so, the on-device check below does not need it
            out["hip_vs_oracle_256"] = "not run: %s" % str(ex)[:120]
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    sims = []
    for schedule in ("passes", "fused"):
        sim = fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=device, schedule=schedule, random=fluid_hip.mulberry32(1234),
                                 storage=storage)
        sims.append(sim)
        sim.multipleSplats(20)
    for sim in sims:
        sim.step(DT, steps)
    # Compared ON THE DEVICE through FluidSim.device_view, which orders torch's current stream behind everything the solver has enqueued —
    # the steps above AND the conversion of the packed dye back to RGBA that asking for the pointer triggers (fluid_stream_wait_context;
    # include/fluid_hip.h, fluid_field_device_ptr's ordering rule).  No host sync is needed in front, none is made.  Round 4 synchronised
    # FIRST and then read the dye while that conversion was still writing it: BENCH_r04's MISMATCH (profiles/r05/device_view_race.txt).
    fields, same = {}, True
    for k in ("velocity", "pressure", "divergence", "curl", "dye"):
        a, b = sims[0].device_view(k), sims[1].device_view(k)
        eq = bool(torch.equal(a, b))
        rec = {"equal": eq}
        if not eq:   # say WHAT differs: a record that only says MISMATCH cannot be diagnosed
            ne = a != b
            rec["n_diff"] = int(ne.sum().item())
            rec["n_values"] = int(a.numel())
            rec["max_abs"] = float((a.double() - b.double()).abs().nan_to_num(nan=float("inf")).max().item())
            rec["n_nan"] = int((a != a).sum().item()) + int((b != b).sum().item())
            idx = ne.nonzero()[0].tolist()
            rec["first_diff_at"] = idx
            rec["first_values"] = [float(a[tuple(idx)].item()), float(b[tuple(idx)].item())]
            # second opinion through the host path (fluid_read_field: a synchronous copy on the solver's own stream)
            rec["host_read_equal"] = bool(np.array_equal(sims[0].read(k), sims[1].read(k)))
        fields[k] = rec
        same = same and eq
    torch.cuda.synchronize(device)
    out["fused_vs_passes_%d" % size] = ("bitwise equal, all five fields after %d steps" % steps) if same else "MISMATCH"
    out["fields_%d" % size] = fields
    for sim in sims:
        sim.close()
    out["seconds"] = round(time.perf_counter() - t0, 2)
    out["ok"] = "MISMATCH" not in json.dumps(out)
    return out


def compare_with_single_domain(one, engines, device):
    """every context of a stripe / tile set against the rows x columns it owns of the single domain `one` (a FluidSim over the GLOBAL grid that
    ran the same splats and steps), all five fields, bit for bit, on the device -> {field: {equal, ...}}.  `engines`: HipStripeEngine-like
    objects (info(name) with row0 / rows / col0 / cols, read(name)) — this rank's one context under RCCL, every context of an in-process set
    in tests/test_bench_live.py."""
    import torch
    fields = {}
    for k in ("velocity", "pressure", "divergence", "curl", "dye"):
        ref, rec = one.device_view(k), {"equal": True}
        for e in engines:
            fi = e.info(k)
            mine = torch.from_numpy(e.read(k)).to(ref.device)            # this context's owned block through the host path
            want = ref[fi.row0:fi.row0 + fi.rows, fi.col0:fi.col0 + fi.cols]
            if want.dtype != mine.dtype:
                want = want.to(mine.dtype)   # fp16 storage: the host path widens exactly
            if mine.dim() == 2:
                mine = mine.unsqueeze(-1)
            if not bool(torch.equal(mine, want)):
                ne = mine != want
                rec.update(equal=False, n_diff=int(ne.sum().item()), n_values=int(mine.numel()),
                           max_abs=float((mine.double() - want.double()).abs().nan_to_num(nan=float("inf")).max().item()),
                           first_diff_at=[int(x) for x in ne.nonzero()[0].tolist()], rows=[fi.row0, fi.row0 + fi.rows], cols=[fi.col0, fi.col0 + fi.cols])
                break
        fields[k] = rec
    torch.cuda.synchronize(device)
    return fields


def decomposition_in_run(fluid_hip, make_set, canvas, cfg, device, schedule, storage, steps=3, splats=6):
    """N > 1: what the ranks exchange is right — on the hardware the number was measured on.  A FRESH stripe / tile set of the benchmark's own
    geometry (collective: its own communicator, the same calibrated link model) takes `splats` seeded splats and `steps` steps over RCCL; every
    rank then runs the single domain of the WHOLE grid itself (the global fields fit one MI355X many times over: 8 GB for 4096 x 32768) with
    the same splats and steps and compares the rows x columns it owns, all five fields, bit for bit.  The library's decomposition is bitwise
    invariant on one GPU (tests/test_stripes_gpu.py, test_baseline_sizes.py: in-process sets and rank threads against a stand-in RCCL); this is
    the same statement over real links, where a stale ghost line or a mis-ordered exchange would show and nothing else in the run would."""
    t0 = time.perf_counter()
    st = make_set()
    try:
        st.multipleSplats(splats)
        st.step(DT, steps)
        st.sync()
        st.check_halo()
        with fluid_hip.FluidSim(canvas=canvas, config=cfg, device=device, schedule=schedule, random=fluid_hip.mulberry32(1234), storage=storage) as one:
            one.multipleSplats(splats)
            one.step(DT, steps)
            fields = compare_with_single_domain(one, [st.engine], device)
    finally:
        st.close()
    ok = all(f["equal"] for f in fields.values())
    return {"set_vs_single_domain": ("bitwise equal on this rank's rows, all five fields after %d steps" % steps) if ok else "MISMATCH",
            "fields": fields, "grid": list(canvas), "seconds": round(time.perf_counter() - t0, 2), "ok": ok}


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
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc child runs (HBM bytes per launch, VALU counters)")
    ap.add_argument("--no-steady", action="store_true", help="skip the long (>= 2000 steps) steady-state timing appended to the line")
    ap.add_argument("--no-decomposition-check", action="store_true", help="N > 1: skip the in-run check of a fresh stripe / tile set against the single "
                    "domain of the whole grid (every rank runs the global grid itself for three steps and compares its rows bit for bit)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity check (fused == per-pass schedule at the bench size, HIP == oracle on a small case) that follows the timed sections")
    ap.add_argument("--extras-budget", type=float, default=600.0, help="seconds everything BEHIND the timed section may take in total "
                    "(counter passes, steady timing, CPU baseline); an extra that no longer fits is skipped and says so in the line")
    ap.add_argument("--tiles-x", type=int, default=1, help="N > 1: 2-D decomposition, N // tiles_x row stripes x tiles_x column tiles "
                                                           "(global grid size*tiles_x x size*N/tiles_x); default 1 = row stripes")
    ap.add_argument("--strong", action="store_true", help="N > 1: strong scaling — the global grid stays --size x --size and is cut into N "
                                                          "stripes / tiles (BASELINE configs[3]: --size 8192 --tiles-x 2 on 4 GPUs; configs[4]: "
                                                          "--size 16384 --iters 200 on 8 GPUs); default: weak scaling, --size x --size per GPU")
    ap.add_argument("--extra-config", action="append", default=None, metavar="SIZE,ITERS,TILES_X[,STEPS]",
                    help="N > 1: after the weak-scaling measurement also time this STRONG-scaling configuration (global SIZE^2 cut into N "
                         "stripes, or N/TILES_X x TILES_X tiles) and report it under `extra_configs`.  Default: BASELINE.json's own multi-GPU "
                         "configurations when N matches — N = 4: 8192,50,2 (configs[3]); N = 8: 16384,200,1 (configs[4]); 'none' disables")
    ap.add_argument("--hosted", action="store_true", help="N > 1: drive the passes from Python with torch.distributed send/recv "
                                                          "instead of the native plan + RCCL inside libfluid_hip.so")
    ap.add_argument("--settle-ms", type=float, default=0.0,
                    help="milliseconds of the same workload on a scratch context enqueued directly in front of the warm-up.  DEFAULT 0: `value` is the "
                         "literal W warm-up + K timed steps, as every earlier round and the driver's consistency check read them.  After an idle of "
                         ">= 1 ms the MI355X answers a load step by dropping its shader clock from 2.4 to ~1.8 GHz for a few milliseconds and ramping "
                         "back over ~15 ms (profiles/r04/first_steps.txt); `--steps 20 --warmup 5` is 13 ms of work, all of it inside that dip.  The "
                         "line reports the same window behind 40 ms of load as an EXTRA field (`preloaded_window`), never as `value`.  > 0: the "
                         "load goes in front of the headline itself; the line then says so (config.settle, effective_warmup_steps, cold_start)")
    ap.add_argument("--link-model", default=None, metavar="US,GBPS|calibrate|default",
                    help="N > 1: what one neighbour message costs on this machine's links (latency in us, GB/s; fluid_set_link_model — sizes how "
                         "much compute the driver puts in front of an exchange's arrival).  Default `calibrate`: measured at start-up on the set's "
                         "own links (fluid_comm_calibrate_link); `default`: the library's 20 us, 50 GB/s.  Recorded in config.link_model")
    ap.add_argument("--no-step-marks", action="store_true", help="no events between the timed steps (`timed_window_regime` is then absent): "
                                                                  "the A/B of what the marks cost")
    ap.add_argument("--comm-timeout", type=float, default=120.0, help="N > 1: seconds the communicator set-up and the warm-up steps may take "
                                                                     "before the watchdog reports which stage hung")
    args = ap.parse_args(argv)
    on_cpu = engine_factory is not None   # only tests/test_bench_multi.py: the launcher path on CPU ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher around us: be the launcher (the ranks re-enter here with WORLD_SIZE set)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv, script=os.path.abspath(sys.argv[0]) if on_cpu else None))

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
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True}
    knobs = fluid_knobs()

    def fail(msg, code=2):
        print("bench.py: " + msg, file=sys.stderr)
        if rank == 0:
            os.write(real_stdout, (json.dumps(dict(base, error=msg)) + "\n").encode())
        sys.exit(code)

    if world != args.gpus:
        fail("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if N > 1 and not on_cpu and "FLUID_RCCL_LIB" in knobs:
        # tests/fake_rccl is an in-process stand-in for single-GPU tests: a number measured over it is not a multi-GPU number
        fail("FLUID_RCCL_LIB=%s is set: bench.py measures over the RCCL that torch ships, not over a stand-in — unset it" % knobs["FLUID_RCCL_LIB"])
    if not on_cpu:
        if not torch.cuda.is_available():
            fail("no GPU visible; the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
    dev_sync = (lambda: None) if on_cpu else torch.cuda.synchronize
    cfg = {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters}
    dog = None
    dist = None
    parity = None

    def make_stripes(gw, gh, its, tx):
        from fluid_hip.stripes import StripeSim
        c = dict(cfg, SIM_RESOLUTION=min(gw, gh), DYE_RESOLUTION=min(gw, gh), PRESSURE_ITERATIONS=its)
        kw = dict(engine_factory=engine_factory) if on_cpu else dict(native=not args.hosted, tiles_x=tx, storage=args.storage,
                                                                    reach=min(args.reach, args.halo),
                                                                    link_model=link_arg(args.link_model))
        return StripeSim(canvas=(gw, gh), config=c, halo=args.halo, schedule=args.schedule, random=fluid_hip.mulberry32(1234),
                         device=local_rank, **kw)

    if N == 1:
        sim = fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=local_rank, schedule=args.schedule,
                                 random=fluid_hip.mulberry32(1234), storage=args.storage)
        sim.multipleSplats(20)
        barrier = lambda: None  # noqa: E731
        grid_w, grid_h = size, size
    else:
        import torch.distributed as dist
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
        dog.at("communicator set-up (ncclGetUniqueId on rank 0, broadcast, ncclCommInitRank inside libfluid_hip.so, link calibration: 46 neighbour exchanges)")
        try:
            sim = make_stripes(gw, gh, iters, tx)
        except Exception as ex:   # no silent switch to another driver: say what failed, on every rank, and stop
            dog.stop()
            fail("stripe driver set-up failed on rank %d: %s" % (rank, ex), code=4)
        sim.multipleSplats(20)
        barrier = dist.barrier
        grid_w, grid_h = gw, gh

    # ---- the chip's power state in front of the timed window (see --settle-ms) ----
    scratch, settle = None, None
    if args.settle_ms > 0 and not on_cpu:
        scratch = fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=local_rank, schedule=args.schedule,
                                     random=fluid_hip.mulberry32(4321), storage=args.storage)
        scratch.multipleSplats(20)
        scratch.step(DT, 2)
        scratch.sync()
        t0 = time.perf_counter()
        scratch.step(DT, 4)
        scratch.sync()
        est = max((time.perf_counter() - t0) / 4, 1e-6)
        n_settle = max(1, int(args.settle_ms / 1e3 / est + 0.999))
        settle = {"steps": n_settle, "est_ms": round(1e3 * est * n_settle, 1),
                  "what": "%d steps of the same %dx%d workload on a scratch context, enqueued directly in front of the warm-up and not waited for: "
                          "the warm-up and the timed steps start on a chip that is already under load (no DVFS dip; profiles/r04/first_steps.txt)"
                          % (n_settle, size, size)}

    def preload():
        if scratch is not None:
            scratch.step(DT, settle["steps"])   # asynchronous: the sync in front of the timed steps waits for it together with the warm-up

    def agree(err):
        """N > 1: a failure on ONE rank (a reach violation is counted per rank) must end EVERY rank — the others would sit in the next
        exchange or collective until the RCCL timeout.  Every rank contributes its flag; all of them fail together."""
        if N == 1:
            return err
        flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) > 0 and not err:
            return "another rank failed (see its message on stderr)"
        return err

    def marks_of(sm):
        """the object that owns the native context of this rank (step marks live in libfluid_hip.so), or None (CPU ranks, hosted driver)"""
        if on_cpu:
            return None
        if hasattr(sm, "set_step_marks"):
            return sm
        eng = getattr(sm, "engine", None)
        return eng if (eng is not None and hasattr(eng, "set_step_marks") and getattr(sm, "native", False)) else None

    def measure(sm, warmup, steps, label, fatal=True):
        """`warmup` untimed steps, then EXACTLY `steps` steps bracketed by sync + barrier + sync; MAX over ranks.
        fatal=False (the extra configurations behind the headline measurement): a failure comes back as a string instead of ending the run"""
        def sync():
            sm.sync()
            dev_sync()
        err = None
        if dog:
            dog.at("%s: warm-up, %d steps (the first ghost-row exchanges over RCCL / xGMI)" % (label, warmup))
        try:
            if marks_of(sm) is not None and steps <= 512 and not args.no_step_marks:
                marks_of(sm).set_step_marks(steps)   # events between the steps, nobody waits for them: `timed_window_regime`
            preload()
            sm.step(DT, warmup)   # N > 1, native driver: the plan and its RCCL exchanges run inside libfluid_hip.so
            sync(); barrier(); sync()
            if dog:
                dog.at("%s: the timed %d steps" % (label, steps))
            t0 = time.perf_counter()
            sm.step(DT, steps)
            sync(); barrier(); sync()
            elapsed = time.perf_counter() - t0
            if N > 1:
                sm.check_halo()
        except Exception as ex:   # FluidError from the library, or anything a hosted driver raises: ONE line either way
            err, elapsed = "step failed on rank %d: %s: %s" % (rank, type(ex).__name__, str(ex)[:300]), 0.0
        err = agree(err)
        if err:
            if not fatal:
                return err
            if dog:
                dog.stop()
            fail(err, code=5)
        if N > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def run(k):
        sim.step(DT, k)

    def sync():
        sim.sync()
        dev_sync()

    elapsed = measure(sim, args.warmup, args.steps, "headline")
    regime = None
    if rank == 0 and marks_of(sim) is not None and args.steps <= 512 and not args.no_step_marks:
        per = marks_of(sim).step_marks()
        marks_of(sim).set_step_marks(0)
        if per:
            # device time of each of the K timed steps, in order: the first steps after the (short) warm-up run slower than the steady state
            # (profiles/r04/first_steps.txt names the mechanism); `head_over_tail` = mean of the first quarter / mean of the last quarter
            q = max(1, len(per) // 4)
            regime = {"ms_per_timed_step": [round(x, 4) for x in per], "sum_ms": round(sum(per), 4),
                      "head_over_tail": round((sum(per[:q]) / q) / max(sum(per[-q:]) / q, 1e-9), 4),
                      "source": "events recorded between the steps inside the timed fluid_step_n call (fluid_set_step_marks; nothing waits for them)"}

    steps_per_s = args.steps / elapsed
    glups = grid_w * grid_h * steps_per_s / 1e9
    half = 0.5 if args.storage == "f16" else 1.0
    alg_step_bytes = algorithmic_bytes_per_cell(iters) * grid_w * grid_h * half
    txn = max(1, args.tiles_x)
    if N == 1:
        which = {(4096, 50): "configs[2]", (1024, 50): "configs[1]"}.get((size, iters), "a side size, not a BASELINE config")
        workload = "%s: %dx%d sim = dye grid, %d Jacobi iters/step, dt=%.6f, 20 splats mulberry32(1234), defaults otherwise" % (which, grid_w, grid_h, iters, DT)
    else:
        workload = ("%s scaling of configs[2]'s per-GPU grid: %dx%d global sim = dye grid as %d ranks of %dx%d (halo %d), %d Jacobi iters/step, "
                    "dt=%.6f, 20 splats mulberry32(1234), defaults otherwise"
                    % ("strong" if args.strong else "weak", grid_w, grid_h, N, grid_w // txn, grid_h * txn // N, args.halo, iters, DT))
    out = dict(base)
    out.update({
        "value": round(glups, 4),
        "steps_per_sec": round(steps_per_s, 3),
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "scaling": "strong" if (args.strong and N > 1) else "weak", "vs_baseline": None,
        "dtype": "f32" if args.storage == "f32" else "f32 arithmetic on f16-stored fields (side measurement, not the headline)",
        "data": "synthetic",
        "config": {"workload": workload, "schedule": args.schedule, "storage": args.storage,
                   "parallelism": "single" if N == 1 else ("stripes%d" % N if args.tiles_x <= 1 else "tiles%dx%d" % (N // args.tiles_x, args.tiles_x))},
        # what the reference's pass structure would have to move for this many steps per second (SURVEY 8d's byte model): with
        # temporal blocking this is a speed-up figure, NOT a fraction of the HBM roofline — the bounded fractions are below
        "speedup_vs_pass_structure": {"algorithmic_GBps": round(alg_step_bytes * steps_per_s / 1e9, 1),
                                      "x_hbm_peak": round(alg_step_bytes * steps_per_s / 1e9 / (HBM_PEAK_GBPS * N), 4)},
    })
    # steps of the same workload the chip ran directly in front of the timed window: the warm-up, plus — only with --settle-ms > 0 — the
    # scratch context's load (its sizing steps included).  Equal to `warmup` for the default command.
    out["effective_warmup_steps"] = args.warmup + (settle["steps"] + 6 if settle else 0)
    if settle:
        out["config"]["settle"] = settle
    # FLUID_* variables in the environment.  Only a LAB build (libfluid_hip_probes.so through FLUID_HIP_LIB) reads the tuning knobs; the
    # product library reads none, so on a product build they are listed apart as ignored — the run measured the shipped defaults.
    if not on_cpu:
        flavor = fluid_hip.lib().fluid_build_flavor().decode()
        out["config"]["build_flavor"] = flavor
        plumbing = ("FLUID_HIP_LIB", "FLUID_RCCL_LIB", "FLUID_HIP_NO_TORCH")
        if flavor == "product":
            ignored = {k: v for k, v in knobs.items() if k not in plumbing}
            if ignored:
                out["config"]["knobs_ignored"] = ignored
                print("bench.py: WARNING: %s set, but the product library reads no tuning knob — this run measures the shipped defaults "
                      "(load the lab build with FLUID_HIP_LIB=.../libfluid_hip_probes.so for A/B runs)" % ", ".join(sorted(ignored)), file=sys.stderr)
            knobs = {k: v for k, v in knobs.items() if k in plumbing}
    out["config"]["knobs"] = knobs   # {} = shipped defaults
    if regime:
        out["timed_window_regime"] = regime
    if N == 1 and not on_cpu:
        si = sim.schedule_info(args.steps, DT)
        # which kernels the timed call launched (the library picks by grid size): chained = steps whose advection launch also ran the next
        # step's curl / vorticity / divergence (fluid_step_n below 3072^2 texels); a per-frame fluid_step never chains
        out["config"]["kernels"] = {"jacobi_shape": si["jacobi_shape"], "jacobi_launches_per_step": si["jacobi_launches"],
                                    "jacobi_chained": bool(si["jacobi_chained"]), "gradsub_folded": bool(si["gradsub_folded"]), "chained_steps": si["chained"],
                                    "curl_field_stored_by_steps": si["curl_stores"], "launches_in_timed_call": si["launches"],
                                    "dye_packed_rgb": bool(si["dye_packed"])}
    if on_cpu:
        out["config"]["engine"] = "injected stripe engine on CPU ranks over %s (launcher-path test, not a measurement)" % backend
    if N > 1:
        out["config"]["exchanges_per_step"] = sim.exchanges / max(args.steps + args.warmup, 1)
        out["config"]["driver"] = "native plan + ncclSend/ncclRecv inside libfluid_hip.so" if sim.native else "hosted: torch.distributed batch_isend_irecv"
        if sim.native:
            out["config"]["advect_exchange_rows"] = list(sim.engine.advect_exchange_rows())
            lm = getattr(sim, "link_model", None)   # this rank's; every rank measures its own links
            out["config"]["link_model"] = ({"latency_us": round(lm[0], 2), "GBps": round(lm[1], 2), "source": lm[2]} if lm
                                           else "library default (20 us + bytes / 50 GB/s per neighbour message)")

    # ---- N > 1: BASELINE.json's own multi-GPU configurations, strong scaling, after the weak-scaling measurement (`value` above is
    #      untouched: the driver's scaling curve is computed from it) ----
    if N > 1:
        plans = args.extra_config
        if plans is None:
            plans = {4: ["8192,50,2"], 8: ["16384,200,1"]}.get(N, [])
        plans = [] if plans == ["none"] else plans
        extra = []
        if dog:
            dog.result = out   # from here on a hang or a failure belongs to an extra configuration: the headline number above is kept
        for spec in plans:
            f = [int(x) for x in spec.split(",")]
            esize, eiters, etx = f[0], f[1], max(1, f[2] if len(f) > 2 else 1)
            esteps = f[3] if len(f) > 3 else max(10, min(args.steps, int(2e10 / (float(esize) * esize * (128 + 12 * eiters) / 728.0 / N * 50))))
            name = {(8192, 50, 2): "configs[3]", (16384, 200, 1): "configs[4]"}.get((esize, eiters, etx), "extra")
            label = "%s: %dx%d sim = dye grid, %d Jacobi iters/step, strong scaling over %d ranks as %s" % (
                name, esize, esize, eiters, N, "%d stripes" % N if etx == 1 else "%dx%d tiles" % (N // etx, etx))
            entry = {"config": label, "steps": esteps, "warmup": max(2, esteps // 5)}
            sim.close()
            if dog:
                dog.at("%s: communicator and field set-up" % name)
            try:
                sim = make_stripes(esize, esize, eiters, etx)
                sim.multipleSplats(20)
                problem = None
            except Exception as ex:
                problem = "set-up failed on rank %d: %s" % (rank, str(ex)[:200])
            problem = agree(problem)
            if problem:
                entry["error"] = problem
                extra.append(entry)
                if problem.startswith("set-up failed"):
                    sim = None
                break
            el = measure(sim, entry["warmup"], esteps, name, fatal=False)
            if isinstance(el, str):
                entry["error"] = el
                extra.append(entry)
                break
            sps = esteps / el
            eb = algorithmic_bytes_per_cell(eiters) * float(esize) * esize * half
            entry.update({"value": round(float(esize) * esize * sps / 1e9, 4), "unit": "GLUPS", "steps_per_sec": round(sps, 3),
                          "ms_per_step": round(1e3 * el / esteps, 4), "scaling": "strong",
                          "speedup_vs_pass_structure": {"algorithmic_GBps": round(eb * sps / 1e9, 1),
                                                        "x_hbm_peak": round(eb * sps / 1e9 / (HBM_PEAK_GBPS * N), 4)},
                          "exchanges_per_step": sim.exchanges / max(esteps + entry["warmup"], 1)})
            extra.append(entry)
        if plans:
            out["extra_configs"] = extra
    if dog:
        dog.stop()

    if scratch is not None:
        scratch.sync()
        scratch.close()
        scratch = None

    deadline = Deadline(args.extras_budget)
    skipped = {}

    # ---- the dominant kernel (the Jacobi loop): launch time from HIP events on the solver's own stream, HBM bytes from PMC passes ----
    if rank == 0 and N == 1 and not args.no_profile_pass:
        sim.set_timing(True)
        sim.step(DT, min(args.steps, 20))
        sim.sync()
        tm = sim.timings()
        sim.set_timing(False)
        # launches of the loop that are ONLY Jacobi (the last launch of a step also carries the gradient subtract under the fused
        # schedule: k_jacobi_tb_gs, timed under gradsub_ms)
        plain = tm["jacobi_launches"] - tm.get("folded_launches", 0)
        only_folded = plain <= 0 and tm.get("folded_launches", 0) > 0
        if only_folded:   # iterations <= the tile depth on a small grid: the step's ONE Jacobi launch also carries the gradient subtract
            launches, avg_ms = tm["folded_launches"], tm["gradsub_ms"] / tm["folded_launches"]
        else:
            launches = max(plain, 1)
            avg_ms = tm["jacobi_ms"] / launches
        avg_ms = max(avg_ms, 1e-6)
        alg_launch = 12.0 * iters * size * size * tm["steps"] / max(tm["jacobi_launches"], 1) * half  # 12 B/cell/iteration, SURVEY.md 8(d)
        # the kernel that runs the loop: the temporally blocked register tile, or one launch per iteration under --schedule passes
        if args.schedule == "fused":
            # (k_jacobi_tb_mix: the same tile kernel with smaller tiles for the launch's first and last rows — what large grids run)
            # (k_jacobi_tb_chain: the loop's blocks of ten iterations as ONE launch — 4096-wide grids since round 5; its counters are per
            # dispatch, the line's figures per BLOCK, which is what a launch of the other kernels is: blocks_per_dispatch below)
            cands = ["k_jacobi_tb_h<", "k_jacobi_tb<"] if args.storage == "f16" else ["k_jacobi_tb_chain<", "k_jacobi_tb_mix<", "k_jacobi_tb<", "k_jacobi_tb2<"]
        else:
            cands = ["k_h_jacobi", "k_jacobi"] if args.storage == "f16" else ["k_jacobi"]
        if only_folded and args.schedule == "fused":
            cands = ["k_jacobi_tb_gs_h<"] if args.storage == "f16" else ["k_jacobi_tb2_gs<", "k_jacobi_tb_gs<"]
        kname = cands[0]
        traffic, why = (None, "--no-traffic") if args.no_traffic else collect_traffic(args, deadline, chained=sim.schedule_info(args.steps, DT)["chained"] > 0)
        entry = None
        if traffic:
            for c in cands:
                hit = [(k, v) for k, v in traffic["kernels"].items() if k.startswith(c)]
                if hit:
                    kname, entry = hit[0]
                    break
            if entry is None:
                why = "no %s dispatch in the counter pass" % cands[0]
        per_dispatch = 1
        if entry and kname.startswith("k_jacobi_tb_chain"):
            per_dispatch = max(1, int(round((launches / max(tm["steps"], 1)) / max(entry["launches_per_step"], 1e-9))))
        if entry:
            bytes_launch, source = int(entry["bytes_per_launch"] / per_dispatch), (
                "PMC FETCH_SIZE x2 + WRITE_SIZE, this run (rocprofv3 --pmc, separate passes): the bytes that crossed the L2's FABRIC side — reads served by "
                "the 256 MiB Infinity Cache are counted like reads from HBM (profiles/r06/mall_probe.txt: a warm 192 MiB sweep runs at 7.0 TB/s with "
                "FETCH_SIZE x2 = the bytes requested), and the Jacobi loop's working set (3 x 64 MiB at 4096^2) fits in that cache")
        else:   # the least a launch must move: pressure in, divergence in, pressure out (no apron re-reads counted)
            bytes_launch, source = int(12.0 * size * size * half), "model: compulsory 12 B/texel per launch (PMC pass unavailable: %s)" % why
        achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
        roof = {
            "kernel": kname, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": bytes_launch, "traffic_source": source,
            "attainable": HBM_ATTAINABLE_GBPS, "frac_of_attainable": round(achieved / HBM_ATTAINABLE_GBPS, 4),
            # the same with the bytes a launch MUST move (pressure in, divergence in, pressure out: 12 B/texel; + 16 velocity in / out when the
            # launch carries the gradient subtract) instead of the bytes it did move: apron re-reads do not count as achievement here
            "compulsory_bytes_per_launch": int((28.0 if only_folded else 12.0) * size * size * half),
            "frac_compulsory": round((28.0 if only_folded else 12.0) * size * size * half / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "avg_launch_ms": round(avg_ms, 5), "launches_per_step": launches / max(tm["steps"], 1),
            # > 1: the kernel is ONE dispatch of that many chained blocks of iterations; traffic, avg_launch_ms and launches_per_step are per block
            "blocks_per_dispatch": per_dispatch,
            "iterations_per_launch": iters * tm["steps"] / max(tm["jacobi_launches"], 1),
            "algorithmic_bytes_per_launch": int(alg_launch),
            "algorithmic_GBps": round(alg_launch / (avg_ms * 1e-3) / 1e9, 1),
            "note": "achieved = HBM bytes one launch really moves / its measured duration (bounded by the peak); "
                    "algorithmic_* = the reference's 12 B/cell/iteration for the iterations this launch performs (a speed-up over the "
                    "pass structure, may exceed the peak); bound = the roofline the path is priced against (HBM: no matrix work on it); co_bound = 'valu' when the arithmetic term exceeds the memory term (frac_of_attainable: achieved / the "
                    "guide's measured 6.29 TB/s streaming ceiling) and the arithmetic term (valu.busy_frac: the average SIMD's "
                    "VALU-issuing cycles / the launch's cycles, SQ counters of this run, x the issue cost of the sweep's instruction mix: "
                    "valu.busy_frac_issue_cost)",
        }
        # the two phases of a temporally blocked launch, timed in this run: ONE iteration (the tile's loads and stores: the memory
        # phase) against the full depth (each further iteration is arithmetic on registers)
        if args.schedule == "fused" and iters >= 2:
            kfull = int(round(roof["iterations_per_launch"]))
            t1, tk = time_jacobi_launch(sim, 1), time_jacobi_launch(sim, kfull)
            roof["mem_phase_ms"] = round(t1, 5)
            roof["valu_phase_ms"] = round(max(tk - t1, 0.0) * kfull / max(kfull - 1, 1), 5)
            roof["phase_note"] = ("launch of 1 iteration / extra time of a launch of %d iterations scaled to %d (standalone launches on "
                                  "random-free state; they overlap partly inside a step, so mem + valu > avg_launch_ms is expected)" % (kfull, kfull))
        if not args.no_traffic and args.schedule == "fused":
            valu, vwhy = collect_valu(args, deadline, cands)
            if valu:
                disp_ms = avg_ms * per_dispatch   # the counters are per dispatch
                valu["busy_ms_at_max_clock"] = round(valu["simd_valu_cycles_per_launch"] / 2.4e6, 5)   # 2.4 GHz: a lower bound on the time
                gui = valu.get("gui_active_cycles_per_launch", 0)
                if gui:   # the counter may come summed over the 8 XCDs: its value per millisecond tells (the clock is 1.8-2.4 GHz)
                    inst = 8 if gui / (disp_ms * 1e6) > 4.8 else 1
                    valu["gui_instances_assumed"] = inst
                    valu["busy_frac"] = round(valu["simd_valu_cycles_per_launch"] / (gui / inst), 4)
                    valu["effective_clock_GHz"] = round(gui / inst / (disp_ms * 1e6), 3)
                else:
                    valu["busy_frac"] = round(valu["busy_ms_at_max_clock"] / disp_ms, 4)
                # busy_frac prices every instruction at one 4-cycle issue (the counter ticks once per instruction: cycles_per_inst reads 4.0).
                # On gfx950 only v_fma_f32 costs that; the sweep's own mix — nine v_pk_add/mul_f32 and two v_add_f32_dpp per four texels —
                # costs 1.65 x (tools/micro/valu_rate2.hip: v_pk_* 1.67, *_dpp 1.55 of a v_fma_f32 slot; profiles/r02/advect_experiments.txt).
                # A MODEL on top of the measured count, hence its own field; the bound is decided with it.
                valu["issue_cost_factor_model"] = JACOBI_ISSUE_COST
                valu["busy_frac_issue_cost"] = round(min(valu["busy_frac"] * JACOBI_ISSUE_COST, 1.0), 4)
                roof["valu"] = valu
                # `bound` names the roofline the path is priced against — HBM: there is no matrix work on it.  When the arithmetic term is the
                # larger one the line says so beside it (the chained launch of round 5 keeps the VALU busy while its tiles wait for nothing)
                if valu["busy_frac_issue_cost"] > roof["frac_of_attainable"]:
                    roof["co_bound"] = "valu"
                    # ... which is the issue-cost MODEL's reading.  Measured (round 6, lab probes on the same launch at 4096^2 / 50,
                    # profiles/r06/chain_bounds_probes.txt): the launch without ANY arithmetic is 26 us of 200 shorter — the loop hides its VALU
                    # work behind three memory round trips in a row per tile (poll, loads, drain of the write-through stores) at two workgroups
                    # per CU; at 8192^2, where launches are four times as long, the same probe saves 400 of 920 us (chain_loop_map.txt)
                    roof["co_bound_note"] = ("model (count x issue cost) — the no-arithmetic probe of this launch at 4096^2 saves 13 % of it: "
                                             "profiles/r06/chain_bounds_probes.txt")
            else:
                roof["valu"] = {"busy_frac": None, "why": vwhy}
        out["roofline"] = roof
        if traffic:
            out["step_hbm"] = {"bytes_per_step": traffic["bytes_per_step"], "GBps": round(traffic["bytes_per_step"] * steps_per_s / 1e9, 1),
                               "frac": round(traffic["bytes_per_step"] * steps_per_s / 1e9 / HBM_PEAK_GBPS, 4),
                               "frac_of_attainable": round(traffic["bytes_per_step"] * steps_per_s / 1e9 / HBM_ATTAINABLE_GBPS, 4),
                               # the bytes the fused schedule MUST move per step (no apron re-reads): curl/vorticity/divergence 20 (+ 4 for a
                               # step that stores its curl field), Jacobi 12 per launch, gradient subtract 20, advection 48
                               "compulsory_bytes_per_step": int(compulsory_step_bytes(size, iters, tm, args.steps, out["config"].get("kernels", {}).get("dye_packed_rgb", False)) * half),
                               "frac_compulsory": round(compulsory_step_bytes(size, iters, tm, args.steps, out["config"].get("kernels", {}).get("dye_packed_rgb", False)) * half * steps_per_s / 1e9 / HBM_PEAK_GBPS, 4),
                               "kernels": {k: {"bytes_per_launch": v["bytes_per_launch"], "launches_per_step": v["launches_per_step"]}
                                           for k, v in traffic["kernels"].items()}}
        per_step = {k: round(v / max(tm["steps"], 1), 4) for k, v in tm.items() if k.endswith("_ms")}
        out["pass_ms_per_step"] = per_step
        notes = []
        if tm.get("folded_launches", 0):
            notes.append("gradsub_ms = the last Jacobi launch of the step with the gradient subtract folded in (k_jacobi_tb_gs); jacobi_ms = the other launches")
        if traffic and any(k.startswith("k_advect_cvd") for k in traffic["kernels"]):
            notes.append("advect_dye_ms = the launch that advects AND runs the next step's curl / vorticity / divergence (k_advect_cvd, fluid_step_n below "
                         "3072^2 texels); vorticity_ms = the first step's own launch only")
        if notes:
            out["pass_ms_note"] = "; ".join(notes)

    # ---- the same loop well inside steady clocks: the contract's K steps may be as few as 20 (11 ms), inside the clock ramp ----
    if rank == 0 and N == 1 and not args.no_steady:
        n_long = max(2000, args.steps)
        if deadline.left() > 20:
            sync()
            t0 = time.perf_counter()
            run(n_long)
            sync()
            out["steady_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / n_long, 4)
            out["steady_steps"] = n_long
        else:
            skipped["steady_ms_per_step"] = "extras budget spent"

    # ---- the page's own call pattern: update() calls step(dt) ONCE per frame (script.js:1176-1186) — K calls of fluid_step, back to back ----
    # (`value` times ONE fluid_step_n(K): a call for K steps stores the curl field once and, below 1536^2, works ahead between its steps;
    # a call per step stores the curl every time and pays the call's own host path.)  Beside it: what a frame costs when the page also
    # composites it — render(target) at the reference's shipping sizes (capture 512, bloom 256 x 8, sunrays 196: script.js:59-85, 1296-1419).
    if rank == 0 and N == 1 and not args.no_steady and not on_cpu and hasattr(sim, "step") and deadline.left() > 20:
        try:
            n_frames = max(200, args.steps)
            for _ in range(30):
                sim.step(DT, 1)
            sync()
            t0 = time.perf_counter()
            for _ in range(n_frames):
                sim.step(DT, 1)
            sync()
            per_frame_ms = 1e3 * (time.perf_counter() - t0) / n_frames
            ref_ms = out.get("steady_ms_per_step") or out["ms_per_step"]
            out["per_frame"] = {"ms_per_step": round(per_frame_ms, 4), "calls": n_frames, "ratio_to_batched": round(ref_ms / per_frame_ms, 4),
                                "batched_ms_per_step": ref_ms,
                                "what": "%d calls of fluid_step (one step each, no synchronisation in between) against one fluid_step_n call of the same steps "
                                        "(`steady_ms_per_step` if measured, else `ms_per_step`): the page's update() pattern, script.js:1176-1186" % n_frames}
            if hasattr(sim, "set_curl_output"):
                # the same calls from a host that — like the page — never looks at the curl field outside step() (script.js:1234-1243) and says so
                # (fluid_set_curl_output(ctx, 0), ABI 10): no step stores it
                sim.set_curl_output(False)
                try:
                    for _ in range(30):
                        sim.step(DT, 1)
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(n_frames):
                        sim.step(DT, 1)
                    sync()
                    nc_ms = 1e3 * (time.perf_counter() - t0) / n_frames
                    out["per_frame"]["without_curl_output"] = {"ms_per_step": round(nc_ms, 4), "ratio_to_batched": round(ref_ms / nc_ms, 4)}
                finally:
                    sim.set_curl_output(True)
            if hasattr(sim, "render"):
                sim.render(512, 512)
                sync()
                t0 = time.perf_counter()
                n_r = 50
                import ctypes as _C
                dp = sim._display_params()
                for _ in range(n_r):
                    sim._check(sim._lib.fluid_render(sim._ctx, 512, 512, _C.byref(dp)))
                sync()
                out["render"] = {"ms": round(1e3 * (time.perf_counter() - t0) / n_r, 4), "target": [512, 512], "calls": n_r,
                                 "what": "fluid_render at the reference's shipping display settings on the %dx%d dye field (shading, bloom 256 x 8, sunrays 196, "
                                         "dithering off-default: script.js:59-85), device time per call incl. the RGBA view of a packed dye field; no readback" % (size, size)}
        except Exception as ex:   # an extra: never takes the line away
            skipped["per_frame"] = "%s: %s" % (type(ex).__name__, str(ex)[:160])

    # ---- the same W + K from a cold chip: what `ms_per_step` would read without the load in front of the warm-up ----
    if rank == 0 and N == 1 and settle and not args.no_steady and deadline.left() > 20:
        sync()
        time.sleep(0.3)
        with fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=local_rank, schedule=args.schedule, random=fluid_hip.mulberry32(1234),
                                storage=args.storage) as cold:
            cold.multipleSplats(20)
            if args.steps <= 512:
                cold.set_step_marks(args.steps)
            cold.step(DT, args.warmup)
            cold.sync(); dev_sync()
            t0 = time.perf_counter()
            cold.step(DT, args.steps)
            cold.sync(); dev_sync()
            el = time.perf_counter() - t0
            per = cold.step_marks() if args.steps <= 512 else []
        out["cold_start"] = {"ms_per_step": round(1e3 * el / args.steps, 4), "ms_per_timed_step": [round(x, 4) for x in per],
                             "what": "the same %d warm-up + %d timed steps on a fresh context after 300 ms of idle, nothing in front of the splats: "
                                     "inside the shader-clock dip that follows a load step (config.settle)" % (args.warmup, args.steps)}

    # ---- the same W + K behind 40 ms of load: what the window reads once the chip is out of the clock dip (an extra, never `value`) ----
    if rank == 0 and N == 1 and not settle and not on_cpu and not args.no_steady and deadline.left() > 20:
        sync()
        with fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=local_rank, schedule=args.schedule, random=fluid_hip.mulberry32(4321),
                                storage=args.storage) as load, \
             fluid_hip.FluidSim(canvas=(size, size), config=cfg, device=local_rank, schedule=args.schedule, random=fluid_hip.mulberry32(1234),
                                storage=args.storage) as hot:
            load.multipleSplats(20)
            hot.multipleSplats(20)
            load.step(DT, 2)
            load.sync()
            t0 = time.perf_counter()
            load.step(DT, 4)
            load.sync()
            est = max((time.perf_counter() - t0) / 4, 1e-6)
            n_load = max(1, int(40.0 / 1e3 / est + 0.999))
            dev_sync()
            time.sleep(0.3)
            load.step(DT, n_load)   # asynchronous, on its own stream: the warm-up below starts on a chip that is already under load
            hot.step(DT, args.warmup)
            hot.sync(); load.sync(); dev_sync()
            t0 = time.perf_counter()
            hot.step(DT, args.steps)
            hot.sync(); dev_sync()
            el = time.perf_counter() - t0
        out["preloaded_window"] = {"ms_per_step": round(1e3 * el / args.steps, 4), "steps_per_sec": round(args.steps / el, 2),
                                   "effective_warmup_steps": args.warmup + n_load,
                                   "what": "the same %d warm-up + %d timed steps on a fresh context, with %d steps (40 ms) of the same workload on a scratch "
                                           "context enqueued directly in front of the warm-up: out of the shader-clock dip that follows a load step "
                                           "(profiles/r04/first_steps.txt).  Reported beside `value`, which is the literal W + K window" % (args.warmup, args.steps, n_load)}

    # the in-run parity check, on every rank's own GPU, BEHIND every timed section of this run (see parity_in_run): a mismatch replaces
    # the line by an error
    if not args.no_parity and not on_cpu:
        problem = None
        if dog:
            dog.at("in-run parity check (fused == per-pass schedule on this rank's GPU)")
        try:
            parity = parity_in_run(fluid_hip, size, iters, local_rank, args.storage, with_oracle=(rank == 0))
            if not parity["ok"]:
                problem = "in-run parity check failed on rank %d: %s" % (rank, json.dumps(parity))
        except Exception as ex:
            problem = "in-run parity check could not run on rank %d: %s" % (rank, str(ex)[:200])
        # The decomposition check below is COLLECTIVE (a fresh communicator, steps over RCCL): the ranks enter it together or not at all.  A
        # parity failure is rank-local at this point, so the ranks first agree on whether ANY of them has one; a rank that failed would
        # otherwise go straight to the all_gather below while the others sit in ncclCommInitRank.  `problem` itself stays
        # per rank: the gathered records say whose it was.
        anyone_failed = N > 1 and agree("x" if problem else None) is not None
        if N > 1 and not anyone_failed and not args.no_decomposition_check:
            # ... and what the ranks EXCHANGE: a fresh set of the benchmark's geometry against the single domain of the whole grid, per rank
            if dog:
                dog.at("in-run decomposition check (a fresh set of %d ranks steps over RCCL; every rank compares its rows with the single domain)" % N)
            try:
                cgl = dict(cfg, SIM_RESOLUTION=min(grid_w, grid_h), DYE_RESOLUTION=min(grid_w, grid_h))
                deco = decomposition_in_run(fluid_hip, lambda: make_stripes(grid_w, grid_h, iters, max(1, args.tiles_x)), (grid_w, grid_h), cgl,
                                            local_rank, args.schedule, args.storage)
                parity["decomposition"] = deco
                if not deco["ok"]:
                    problem = "in-run decomposition check failed on rank %d: %s" % (rank, json.dumps(deco)[:1200])
            except Exception as ex:   # the check could not run (memory, set-up): say so; only a MISMATCH is an error of the run
                parity["decomposition"] = {"set_vs_single_domain": "not run: %s: %s" % (type(ex).__name__, str(ex)[:200]), "ok": None}
        if N > 1:   # every rank checks its own GPU: the line carries all of their summaries, so that one rank's problem is attributable
            mine = {"rank": rank, "ok": bool(parity and parity["ok"]) and not problem}
            if parity:
                mine["fields_differing"] = {k: v for k, v in parity.get("fields_%d" % size, {}).items() if not v.get("equal")}
                if "decomposition" in parity:
                    mine["decomposition"] = parity["decomposition"]["set_vs_single_domain"]
            if problem:
                mine["problem"] = problem[:400]
            ranks = [None] * N
            dist.all_gather_object(ranks, mine)
            if parity is None:
                parity = {"ok": False}
            parity["ranks"] = ranks
            bad = [r for r in ranks if not r["ok"]]
            if bad and not problem:
                problem = "in-run parity check failed on rank(s) %s: %s" % ([r["rank"] for r in bad], json.dumps(bad)[:1500])
        problem = agree(problem)
        if problem:
            if dog:
                dog.stop()
            fail(problem, code=6)

    if parity:
        out["parity_in_run"] = parity

    if rank == 0 and N == 1 and args.cpu_budget > 0:
        if deadline.left() > 30:
            out["cpu_baseline"] = cpu_baseline(size, iters, min(args.cpu_budget, deadline.left() / 12.0), args.cpu_kind, max_wait=deadline.left())
        else:
            out["cpu_baseline"] = {"value": None, "unit": "GLUPS", "kind": "reference", "cores": os.cpu_count(),
                                   "sample": "not measured: the extras budget (--extras-budget) was spent by the counter passes"}
    if skipped:
        out["skipped"] = skipped

    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def compulsory_step_bytes(size, iters, tm, steps_in_call, dye_packed=False):
    """HBM bytes one fused step cannot avoid at `size`^2 (fp32, dye grid = sim grid), from the launches the timing pass counted"""
    per_step_launches = tm["jacobi_launches"] / max(tm["steps"], 1)
    folded = tm.get("folded_launches", 0) / max(tm["steps"], 1)
    curl = 4.0 / max(steps_in_call, 1)   # only the call's last step stores the curl field
    b = 20.0 + curl + 12.0 * per_step_launches + (16.0 if folded else 20.0) + (40.0 if dye_packed else 48.0)   # advection: velocity 8 + 8, dye 16 + 16 (12 + 12 packed)
    return b * size * size


def time_jacobi_launch(sim, k, reps=20):
    """milliseconds of one standalone temporally blocked launch of k iterations on the bench's own fields (fluid_pass_jacobi)"""
    for _ in range(3):
        sim.run_pass("jacobi", iters=k)
    sim.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        sim.run_pass("jacobi", iters=k)
    sim.sync()
    return 1e3 * (time.perf_counter() - t0) / reps


if __name__ == "__main__":
    main()
