"""TEST INFRASTRUCTURE — times the UNMODIFIED reference step() (script.js:1231-1294) under Chromium + SwiftShader (software
WebGL, from the kaleido package) on this host's cores, with the scenario bench.py runs on the GPU: zero state,
multipleSplats(20) from mulberry32(1234), dt = 0.016666, per-step readPixels sync.

  python oracle/live/time_reference.py                      the three BASELINE sizes -> oracle/live/reference_timing.json
  python oracle/live/time_reference.py --size 4096 --iters 50 --warm 3 --timed 5 --json
                                                             one size, ONE JSON line on stdout (bench.py's `cpu_baseline` leg)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import live_reference as live  # noqa: E402


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_one(size, iters, warm, timed):
    r = live.run({"canvasW": 512, "canvasH": 512, "seed": 1234, "randomSplats": 20, "timing": True, "noDump": True,
                  "config": {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters},
                  "steps": warm + timed})
    ms = r["ms"][warm:]
    mean = sum(ms) / len(ms)
    return {"ms_per_step": round(mean, 2), "steps_per_sec": round(1e3 / mean, 4), "GLUPS": round(size * size / mean / 1e6, 5),
            "warmup_steps": warm, "timed_steps": len(ms), "ms": [round(x, 1) for x in r["ms"]], "sim": r["sim"], "dye": r["dye"],
            "gl": r["gl"], "nproc": os.cpu_count(), "cpu_model": cpu_model(), "reference_dir": live.REFERENCE_DIR}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--timed", type=int, default=5)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    if a.size:
        out = time_one(a.size, a.iters, a.warm, a.timed)
        print(json.dumps(out) if a.json else out, flush=True)
        return
    out = {}
    for size, iters, warm, timed in ((128, 20, 5, 40), (1024, 50, 3, 20), (4096, 50, 3, 5)):
        out["%d^2/%d" % (size, iters)] = time_one(size, iters, warm, timed)
        print(size, iters, out["%d^2/%d" % (size, iters)], flush=True)
    out["nproc"] = os.cpu_count()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
