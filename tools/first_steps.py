#!/usr/bin/env python3
"""GPU box: why do the first ~25 steps after the splats run slower than the steady state?

The driver times `bench.py --steps 20 --warmup 5`: steps 6..25 after the splats, 11 ms of GPU work.  This tool separates the candidate
mechanisms on the bench workload (4096^2 / 50, 20 seeded splats), everything on ONE stream with no host sync inside a phase:

  * per-step device time from events recorded between the steps (no sync: the chip stays loaded);
  * the effective SHADER clock at every step boundary, measured from inside the stream by tools/micro/clock_probe.hip
    (s_memtime ticks / s_memrealtime: a 3 us one-wave kernel) — rocm-smi samples once a second and cannot see a ramp of milliseconds;
  * what the driver exposes about memory / fabric clocks and power, sampled by a side thread as fast as sysfs answers
    (pp_dpm_sclk / mclk / fclk, hwmon power and freq) — whichever of those files exist on the box;
  * first touch / TLB against DVFS: the same 60 steps again on the SAME (already touched) buffers after idles of 200 / 20 / 5 / 1 / 0.2 ms,
    and right after a bare sync;
  * prior load against none: 100 steps of ANOTHER context on the same stream directly in front of the splats + 25 steps of a fresh one.

Usage: python tools/first_steps.py [--size 4096] [--iters 50] > profiles/r04/first_steps_raw.txt
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "webgl-fluid-simulation_amd"))
DT = 0.016666


class Sampler:
    """reads whatever clock / power files the amdgpu driver exposes, as fast as they answer, with host timestamps"""

    def __init__(self):
        self.files = {}
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
                f = os.path.join(card, name)
                if os.path.exists(f):
                    self.files[name] = f
            for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input"):
                    f = os.path.join(hw, name)
                    if os.path.exists(f):
                        self.files[name] = f
            if self.files:
                break
        self.rows = []
        self.stop = threading.Event()
        self.t = None

    @staticmethod
    def _parse(name, text):
        if name.startswith("pp_dpm"):
            for line in text.splitlines():   # "1: 2100Mhz *" marks the active level
                if line.rstrip().endswith("*"):
                    return line.split(":")[1].replace("*", "").strip()
            return text.strip().splitlines()[-1] if text.strip() else None
        return text.strip()

    def _run(self):
        while not self.stop.is_set():
            row = {"t": time.perf_counter()}
            for name, f in self.files.items():
                try:
                    with open(f) as fh:
                        row[name] = self._parse(name, fh.read())
                except OSError:
                    row[name] = None
            self.rows.append(row)

    def start(self):
        if self.files:
            self.t = threading.Thread(target=self._run, daemon=True)
            self.t.start()

    def finish(self):
        self.stop.set()
        if self.t:
            self.t.join(2.0)

    def between(self, t0, t1):
        return [r for r in self.rows if t0 <= r["t"] <= t1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--chain", type=int, default=500, help="length of the probe kernel's dependent chain (x4 v_fma_f32)")
    args = ap.parse_args()
    import torch
    import fluid_hip
    lib = fluid_hip.lib()
    probe = C.CDLL(os.path.join(ROOT, "tools", "micro", "libclock_probe.so"))
    probe.clock_probe_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    N = args.size
    cfg = {"SIM_RESOLUTION": N, "DYE_RESOLUTION": N, "PRESSURE_ITERATIONS": args.iters}
    stream = torch.cuda.Stream()
    sptr = C.c_void_p(stream.cuda_stream)
    MAXS = 4096
    buf = torch.zeros(MAXS * 4, dtype=torch.int64, device="cuda")
    slot = [0]

    def make_sim(seed=1234):
        sim = fluid_hip.FluidSim(canvas=(N, N), config=cfg, random=fluid_hip.mulberry32(seed))
        assert lib.fluid_set_stream(sim._ctx, sptr, 1) == 0
        return sim

    def phase(sim, n, label, before=None):
        """n single steps back to back on the stream: events and clock probes between them; returns the rows"""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        first = slot[0]
        t_host0 = time.perf_counter()
        if before:
            before()
        evs[0].record(stream)
        for k in range(n):
            sim.step(DT, 1)
            probe.clock_probe_launch(sptr, C.c_void_p(buf.data_ptr()), slot[0], args.chain)
            slot[0] += 1
            evs[k + 1].record(stream)
        stream.synchronize()
        t_host1 = time.perf_counter()
        raw = buf[4 * first: 4 * (first + n)].cpu().numpy().reshape(n, 4)
        rows = []
        for k in range(n):
            ms = evs[k].elapsed_time(evs[k + 1])
            ticks, real = int(raw[k, 0]), int(raw[k, 1])
            rows.append((k + 1, ms, ticks / max(real, 1) * 100.0))   # MHz = ticks per 10 ns x 100
        print("## %s   (host wall %.2f ms for %d steps)" % (label, 1e3 * (t_host1 - t_host0), n))
        return rows, (t_host0, t_host1)

    def show(rows, groups=((1, 1), (2, 2), (3, 3), (4, 5), (6, 10), (11, 15), (16, 25), (26, 40), (41, 60), (61, 100), (101, 200), (201, 400))):
        for a, b in groups:
            sel = [r for r in rows if a <= r[0] <= b]
            if not sel:
                continue
            ms = sum(r[1] for r in sel) / len(sel)
            mhz = sum(r[2] for r in sel) / len(sel)
            print("   steps %3d..%3d  %.4f ms/step   shader clock %6.0f MHz (min %6.0f)" % (a, b, ms, mhz, min(r[2] for r in sel)))
        sys.stdout.flush()

    def window(rows, a, b):
        sel = [r for r in rows if a <= r[0] <= b]
        return sum(r[1] for r in sel) / max(len(sel), 1)

    smp = Sampler()
    print("# sysfs files sampled: %s" % (json.dumps(smp.files) if smp.files else "none visible in this container"))
    smp.start()
    torch.cuda.synchronize()
    time.sleep(0.5)

    def sys_summary(t0, t1, label):
        rows = smp.between(t0, t1)
        if not rows:
            return
        keys = [k for k in rows[0] if k != "t"]
        seq = []
        last = None
        for r in rows:
            v = tuple(r.get(k) for k in keys)
            if v != last:
                seq.append(("%.2f ms" % (1e3 * (r["t"] - t0)),) + v)
                last = v
        print("   sysfs during %s (%d samples, changes only; columns: t, %s):" % (label, len(rows), ", ".join(keys)))
        for s in seq[:40]:
            print("      " + "  ".join(str(x) for x in s))

    with torch.cuda.stream(stream):
        # ---- A: the bench's own sequence: fresh context, 20 splats, then steps ----
        sim = make_sim()
        sim.multipleSplats(20)
        rows, (t0, t1) = phase(sim, 60, "A  fresh context, 20 splats, then 60 single steps (the driver times steps 6..25)")
        show(rows)
        print("   driver window (steps 6..25): %.4f ms/step;  steps 41..60: %.4f" % (window(rows, 6, 25), window(rows, 41, 60)))
        sys_summary(t0, t1, "A")
        steady, _ = phase(sim, 300, "A' 300 more steps without a pause (the steady state of this box)")
        show(steady, groups=((1, 25), (26, 100), (101, 200), (201, 300)))
        ref = window(steady, 101, 300)
        print("   steady reference: %.4f ms/step" % ref)

        # ---- B: the same buffers (all touched, TLB warm as far as it gets), after an idle: DVFS / power state only ----
        for idle_ms in (200.0, 20.0, 5.0, 1.0, 0.2, 0.0):
            stream.synchronize()
            if idle_ms:
                time.sleep(idle_ms / 1e3)
            rows, (t0, t1) = phase(sim, 40, "B  touched buffers, after a sync + %.1f ms idle: 40 steps" % idle_ms)
            show(rows, groups=((1, 1), (2, 2), (3, 5), (6, 10), (11, 25), (26, 40)))
            print("   steps 6..25 of this phase: %.4f ms/step = %+.1f %% against the steady %.4f" % (window(rows, 6, 25), 100 * (window(rows, 6, 25) / ref - 1), ref))
            if idle_ms in (200.0, 5.0):
                sys_summary(t0, t1, "B idle %.0f ms" % idle_ms)

        # ---- C: a FRESH context (untouched buffers) straight behind sustained load on another context: first touch without the DVFS part ----
        sim2 = make_sim()
        rows, _ = phase(sim2, 40, "C  fresh context behind 100 steps of the first one on the same stream, no idle: splats + 40 steps",
                        before=lambda: (sim.step(DT, 100), sim2.multipleSplats(20)))
        show(rows, groups=((1, 1), (2, 2), (3, 5), (6, 10), (11, 25), (26, 40)))
        print("   steps 6..25: %.4f ms/step (A read %s for the same flow state)" % (window(rows, 6, 25), "above"))
        sim2.close()

        # ---- D: a fresh context after an idle, as A, for the box-to-box / repeat spread ----
        stream.synchronize()
        time.sleep(0.3)
        sim3 = make_sim()
        sim3.multipleSplats(20)
        rows, (t0, t1) = phase(sim3, 40, "D  = A again (fresh context after 300 ms idle)")
        show(rows, groups=((1, 1), (2, 2), (3, 5), (6, 10), (11, 25), (26, 40)))
        print("   steps 6..25: %.4f ms/step" % window(rows, 6, 25))
        sim3.close()

        # ---- E: per-pass view of the first steps (library timing mode: a sync per pass, so the chip idles between passes; shape only) ----
        stream.synchronize()
        time.sleep(0.3)
        sim4 = make_sim()
        sim4.multipleSplats(20)
        print("## E  per-pass device times, fresh context (timing mode: one sync per pass group)")
        for a, n in ((1, 5), (6, 20), (26, 40), (66, 100)):
            sim4.set_timing(True)
            sim4.step(DT, n)
            sim4.sync()
            tm = sim4.timings()
            sim4.set_timing(False)
            print("   steps %3d..%3d  %s" % (a, a + n - 1, {k[:-3]: round(1e3 * v / n, 1) for k, v in tm.items() if k.endswith("_ms") and v}))
        sim4.close()
        sim.close()
    smp.finish()
    print("# sysfs sampler: %d samples in total, %.0f per second" % (len(smp.rows), len(smp.rows) / max(smp.rows[-1]["t"] - smp.rows[0]["t"], 1e-9) if smp.rows else 0))


if __name__ == "__main__":
    main()
