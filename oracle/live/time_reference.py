"""TEST INFRASTRUCTURE — times the UNMODIFIED reference step() under Chromium/SwiftShader (software WebGL)
on this container's cores, with the same scenario bench.py uses on the GPU.
Build-container only (needs /root/reference).  Output is quoted in DESIGN.md / BASELINE notes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import live_reference as live  # noqa: E402


def main():
    out = {}
    for size, iters, warm, timed in ((128, 20, 5, 40), (1024, 50, 3, 20), (4096, 50, 3, 5)):
        r = live.run({"canvasW": 512, "canvasH": 512, "seed": 1234, "randomSplats": 20, "timing": True, "noDump": True,
                      "config": {"SIM_RESOLUTION": size, "DYE_RESOLUTION": size, "PRESSURE_ITERATIONS": iters},
                      "steps": warm + timed})
        ms = r["ms"][warm:]
        mean = sum(ms) / len(ms)
        out["%d^2/%d" % (size, iters)] = {"ms_per_step": round(mean, 2), "steps_per_sec": round(1e3 / mean, 4),
                                          "GLUPS": round(size * size / mean / 1e6, 5), "timed_steps": len(ms), "gl": r["gl"]}
        print(size, iters, out["%d^2/%d" % (size, iters)], flush=True)
    out["nproc"] = os.cpu_count()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
