"""TEST INFRASTRUCTURE — drives the UNMODIFIED reference (`/root/reference/script.js`)
headless under Chromium 88 + SwiftShader (software WebGL2) as shipped inside the `kaleido`
pip package, and returns fp32 dumps of its simulation fields.

Needs the reference's script.js: `/root/reference` (the build container) or the staged copy `oracle/_ref/`
(oracle/stage_reference.sh; that is what bench.py's `cpu_baseline` leg finds on the GPU box).  Used by
`oracle/live/make_golden.py` to generate `tests/golden/*.npz` and by
`oracle/live/time_reference.py` for the reference-side timing quoted in DESIGN.md.
`bench.py` imports it for its `cpu_baseline` leg only (the reference timed on the host cores, beside the GPU number);
nothing under `tests/ -m gpu` or `smoke()` does.

Protocol: kaleido reads one JSON request per line on stdin and answers one JSON line on
stdout whose "result" is the string our fake `Plotly.toImage` resolved (oracle_plotly.js).
"""
from __future__ import annotations

import base64
import json
import os
import subprocess
from typing import Any, Dict, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# the reference's page script: the read-only checkout in the build container, or the byte-for-byte staged copy that
# oracle/stage_reference.sh puts under oracle/_ref/ (git-ignored; travels to the GPU box with the gpurun snapshot)
_CANDIDATES = tuple(filter(None, (os.environ.get("FLUID_REFERENCE_DIR"), "/root/reference", os.path.join(os.path.dirname(HERE), "_ref"))))
REFERENCE_DIR = next((d for d in _CANDIDATES if os.path.exists(os.path.join(d, "script.js"))), _CANDIDATES[0])


def _kaleido_exe() -> Optional[str]:
    try:
        import kaleido  # noqa: F401
    except Exception:
        return None
    exe = os.path.join(os.path.dirname(kaleido.__file__), "executable", "kaleido")
    return exe if os.path.exists(exe) else None


def available() -> bool:
    return _kaleido_exe() is not None and os.path.exists(os.path.join(REFERENCE_DIR, "script.js"))


def _b64(a: np.ndarray) -> str:
    return base64.b64encode(np.ascontiguousarray(a, dtype=np.float32).tobytes()).decode()


def run(scenario: Dict[str, Any], timeout: float = 1800.0) -> Dict[str, Any]:
    """Run one scenario in a fresh browser process. Returns the decoded reply;
    reply["fields"][name] is a float32 array [H, W, 4] (row 0 = bottom)."""
    exe = _kaleido_exe()
    if exe is None or not available():
        raise RuntimeError("live reference unavailable (needs kaleido and /root/reference or oracle/_ref)")
    sc = dict(scenario)
    sc.setdefault("refdir", "file://" + REFERENCE_DIR.rstrip("/") + "/")
    inj = sc.get("inject")
    if inj:
        sc["inject"] = {k: _b64(v) for k, v in inj.items()}
    args = [exe, "plotly", "--plotlyjs=file://" + os.path.join(HERE, "oracle_plotly.js"), "--no-sandbox",
            "--allow-file-access-from-files", "--disable-breakpad", "--disable-dev-shm-usage", "--disable-gpu"]
    p = subprocess.Popen(args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        p.stdout.readline()  # startup banner
        req = {"data": {"data": [], "layout": sc}, "format": "svg", "width": 100, "height": 100, "scale": 1}
        p.stdin.write((json.dumps(req) + "\n").encode())
        p.stdin.flush()
        line = p.stdout.readline()
    finally:
        try:
            p.stdin.close()
        except Exception:
            pass
        try:
            p.wait(timeout=10)
        except Exception:
            p.kill()
    outer = json.loads(line)
    if outer.get("code", 0) != 0 or "result" not in outer:
        raise RuntimeError("kaleido error: %r" % (outer,))
    res = json.loads(outer["result"])
    if "error" in res:
        raise RuntimeError("reference harness error: " + res["error"])
    if "fields" in res:
        (sw, sh), (dw, dh) = res["sim"], res["dye"]
        dec = {}
        for name, s in res["fields"].items():
            w, h = (dw, dh) if name == "dye" else (sw, sh)
            dec[name] = np.frombuffer(base64.b64decode(s), np.float32).reshape(h, w, 4).copy()
        res["fields"] = dec
    if "samples" in res:   # subsampled dump (oracle_plotly.js `sample`): native channel counts
        (sw, sh), (dw, dh) = res["sim"], res["dye"]
        nch = {"velocity": 2, "pressure": 1, "divergence": 1, "curl": 1, "dye": 4}
        dec = {}
        for name, smp in res["samples"].items():
            w = dw if name == "dye" else sw
            nw, nh = smp["size"]
            sub = np.frombuffer(base64.b64decode(smp["sub"]), np.float32).reshape(nh, nw, 4)[..., :nch[name]]
            band = np.frombuffer(base64.b64decode(smp["band"]), np.float32).reshape(-1, w, 4)[..., :nch[name]]
            if nch[name] == 1:
                sub, band = sub[..., 0], band[..., 0]
            dec[name] = {"sub": np.ascontiguousarray(sub), "band": np.ascontiguousarray(band), "absmax": float(smp["absmax"])}
        res["samples"] = dec
    if "frame" in res:   # render scenario: float frame, 8-bit frame (already flipped by normalizeTexture), bloom, sunrays, mask
        fw, fh = res["frameSize"]
        res["frame"] = np.frombuffer(base64.b64decode(res["frame"]), np.float32).reshape(fh, fw, 4).copy()
        res["frame8"] = np.frombuffer(base64.b64decode(res["frame8"]), np.uint8).reshape(fh, fw, 4).copy()
        bw, bh = res["bloomSize"]
        res["bloom"] = np.frombuffer(base64.b64decode(res["bloom"]), np.float32).reshape(bh, bw, 4).copy()
        sw_, sh_ = res["sunraysSize"]
        res["sunrays"] = np.frombuffer(base64.b64decode(res["sunrays"]), np.float32).reshape(sh_, sw_, 4)[..., 0].copy()
        dw_, dh_ = res["dye"]
        res["mask"] = np.frombuffer(base64.b64decode(res["mask"]), np.float32).reshape(dh_, dw_, 4).copy()
    return res


def native_channels(fields: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Strip readPixels' RGBA padding (script.js:301-307): velocity RG, scalars R, dye RGBA."""
    return {
        "velocity": np.ascontiguousarray(fields["velocity"][..., :2]),
        "pressure": np.ascontiguousarray(fields["pressure"][..., 0]),
        "divergence": np.ascontiguousarray(fields["divergence"][..., 0]),
        "curl": np.ascontiguousarray(fields["curl"][..., 0]),
        "dye": np.ascontiguousarray(fields["dye"]),
    }


if __name__ == "__main__":
    r = run({"canvasW": 256, "canvasH": 256, "config": {"SIM_RESOLUTION": 32, "DYE_RESOLUTION": 32},
             "splats": [[0.5, 0.5, 300.0, -200.0, 1.0, 0.5, 0.25]], "steps": 1})
    print({k: v for k, v in r.items() if k != "fields"})
    print({k: (v.shape, float(np.abs(v).max())) for k, v in r["fields"].items()})
