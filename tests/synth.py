"""Synthetic full-size inputs that the live-reference harness generates IN THE PAGE (oracle/live/oracle_plotly.js `synth`) and the
tests regenerate here bit for bit: a compact vortex (polynomial in doubles) plus mulberry32 white noise, rounded to fp32 once.
Only + - * / max on IEEE doubles on both sides, so JavaScript and numpy agree on every bit."""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _imul(a, b):
    return (a * b) & M32   # low 32 bits of the product (uint64 arithmetic on values < 2^32)


def mulberry32_block(seed: int, start: int, n: int) -> np.ndarray:
    """Draws start+1 .. start+n of mulberry32(seed) as float64 in [0, 1): the generator's state is seed + k * 0x6D2B79F5."""
    k = np.arange(start + 1, start + n + 1, dtype=np.uint64)
    st = (np.uint64(seed & 0xFFFFFFFF) + k * np.uint64(0x6D2B79F5)) & M32
    t = _imul(st ^ (st >> np.uint64(15)), np.uint64(1) | st)
    t = ((t + _imul(t ^ (t >> np.uint64(7)), np.uint64(61) | t)) & M32) ^ t
    r = (t ^ (t >> np.uint64(14))) & M32
    return r.astype(np.float64) / 4294967296.0


def field(w: int, h: int, nch: int, spec: dict, rows=None) -> np.ndarray:
    """The field the page uploads for `spec` = {seed, noise, amp, cx, cy, R2}; [h, w, nch] float32 (or [h, w] for nch == 1).
    `rows` = (r0, r1) generates that row range only."""
    r0, r1 = rows if rows else (0, h)
    amp = list(spec.get("amp", [0, 0, 0, 0])) + [0, 0, 0, 0]
    cx, cy, R2, noise = spec.get("cx", 0.5), spec.get("cy", 0.5), spec.get("R2", 0.1), spec.get("noise", 0)
    out = np.empty((r1 - r0, w, nch), np.float32)
    x = (np.arange(w, dtype=np.float64) + 0.5) / w
    dx = x - cx
    CH = 256   # rows per block (bounds the float64 temporaries)
    for a in range(r0, r1, CH):
        b = min(a + CH, r1)
        y = ((np.arange(a, b, dtype=np.float64) + 0.5) / h)[:, None]
        dy = y - cy
        r2 = dx[None, :] * dx[None, :] + dy * dy
        g = np.maximum(0.0, 1.0 - r2 / R2)
        g = g * g
        rnd = mulberry32_block(spec["seed"], a * w * nch, (b - a) * w * nch).reshape(b - a, w, nch)
        for c in range(nch):
            sh = (-dy * g) if c == 0 else (dx[None, :] * g) if c == 1 else g if c == 2 else (x[None, :] * y)
            out[a - r0:b - r0, :, c] = (amp[c] * sh + (rnd[..., c] - 0.5) * noise).astype(np.float32)
    return out[..., 0] if nch == 1 else out
