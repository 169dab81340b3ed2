"""TEST INFRASTRUCTURE — CPU restatement (numpy, fp32) of the reference's display compositor: render(target) with
bloom, sunrays, shading and the back-colour blend, as captureScreenshot() drives it (script.js:287-349, 1296-1419;
shaders 520-724, 460-494).  The checker for the HIP display path; never shipped, never on a product
path.  Pinned to the live reference by tests/golden/display_*.npz (oracle/live/make_golden_display.py).

Conventions as in oracle.py: arrays [H, W, C], row 0 = bottom; vUv of a target texel = ((i + .5) / w, (j + .5) / h)
(baseVertexShader, script.js:440-459); every FBO texture is LINEAR + CLAMP_TO_EDGE (script.js:1051-1052, 1016, 1036),
the dithering texture LINEAR + REPEAT (script.js:1131-1134)."""
from __future__ import annotations

import numpy as np

F = np.float32


def _grid(w: int, h: int):
    u = ((np.arange(w, dtype=F) + F(0.5)) / F(w))[None, :].repeat(h, 0)
    v = ((np.arange(h, dtype=F) + F(0.5)) / F(h))[:, None].repeat(w, 1)
    return u.astype(F), v.astype(F)


def sample(tex: np.ndarray, u: np.ndarray, v: np.ndarray, repeat: bool = False) -> np.ndarray:
    """texture2D with LINEAR filtering; mix(a, b, t) = a + (b - a) * t, the form validated against SwiftShader"""
    H, W = tex.shape[:2]
    x = (u * F(W) - F(0.5)).astype(F)
    y = (v * F(H) - F(0.5)).astype(F)
    i0 = np.floor(x)
    j0 = np.floor(y)
    fx = (x - i0).astype(F)
    fy = (y - j0).astype(F)
    i0 = i0.astype(np.int64)
    j0 = j0.astype(np.int64)
    if repeat:
        ia, ib, ja, jb = i0 % W, (i0 + 1) % W, j0 % H, (j0 + 1) % H
    else:
        ia, ib = np.clip(i0, 0, W - 1), np.clip(i0 + 1, 0, W - 1)
        ja, jb = np.clip(j0, 0, H - 1), np.clip(j0 + 1, 0, H - 1)
    a, b, c, d = tex[ja, ia], tex[ja, ib], tex[jb, ia], tex[jb, ib]
    if tex.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    lo = (a + (b - a) * fx).astype(F)
    hi = (c + (d - c) * fx).astype(F)
    return (lo + (hi - lo) * fy).astype(F)


# ---- bloom: script.js:1346-1389, shaders 614-675 -------------------------------------------------------------------
def bloom_prefilter(dye: np.ndarray, bw: int, bh: int, threshold: float, soft_knee: float) -> np.ndarray:
    knee = threshold * soft_knee + 0.0001                      # JS doubles, narrowed by gl.uniform3f / uniform1f
    c0, c1, c2, th = F(threshold - knee), F(knee * 2), F(0.25 / knee), F(threshold)
    u, v = _grid(bw, bh)
    c = sample(dye, u, v)[..., :3]
    br = np.maximum(c[..., 0], np.maximum(c[..., 1], c[..., 2]))
    rq = np.clip(br - c0, F(0), c1).astype(F)
    rq = (c2 * rq * rq).astype(F)
    k = (np.maximum(rq, br - th) / np.maximum(br, F(0.0001))).astype(F)
    out = np.zeros((bh, bw, 4), F)
    out[..., :3] = c * k[..., None]
    return out


def box4(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """bloomBlurShader / bloomFinalShader body: 0.25 * (L + R + T + B), taps one SOURCE texel away from vUv"""
    H, W = src.shape[:2]
    tx, ty = F(1.0 / W), F(1.0 / H)
    u, v = _grid(dw, dh)
    s = sample(src, u - tx, v)
    s = s + sample(src, u + tx, v)
    s = s + sample(src, u, v + ty)
    s = s + sample(src, u, v - ty)
    return (s * F(0.25)).astype(F)


def bloom_levels(bw: int, bh: int, iterations: int):
    out = []
    for i in range(iterations):
        w, h = bw >> (i + 1), bh >> (i + 1)
        if w < 2 or h < 2:
            break
        out.append((w, h))
    return out


def apply_bloom(dye, bw, bh, iterations, intensity, threshold, soft_knee, previous=None):
    levels = bloom_levels(bw, bh, iterations)
    if len(levels) < 2:                                         # script.js:1347-1348: bloom keeps whatever it held
        return previous if previous is not None else np.concatenate([np.zeros((bh, bw, 3), F), np.ones((bh, bw, 1), F)], -1)
    last = bloom_prefilter(dye, bw, bh, threshold, soft_knee)
    bufs = []
    for (w, h) in levels:
        last = box4(last, w, h)
        bufs.append(last)
    for i in range(len(bufs) - 2, -1, -1):                      # blendFunc(ONE, ONE): dst = src + dst
        h, w = bufs[i].shape[:2]
        bufs[i] = (box4(last, w, h) + bufs[i]).astype(F)
        last = bufs[i]
    return (box4(last, bw, bh) * F(intensity)).astype(F)


# ---- sunrays: script.js:1391-1403, shaders 677-724; blur 1405-1419, shaders 460-494 ---------------------------------
def sunrays_mask(dye: np.ndarray) -> np.ndarray:
    out = dye.astype(F).copy()
    br = np.maximum(dye[..., 0], np.maximum(dye[..., 1], dye[..., 2]))
    out[..., 3] = F(1.0) - np.minimum(np.maximum(br * F(20.0), F(0.0)), F(0.8))
    return out


def sunrays_march(mask: np.ndarray, sw: int, sh: int, weight: float) -> np.ndarray:
    a = np.ascontiguousarray(mask[..., 3])
    u, v = _grid(sw, sh)
    k = F(F(1.0) / F(16.0) * F(0.3))
    du, dv = ((u - F(0.5)) * k).astype(F), ((v - F(0.5)) * k).astype(F)
    cu, cv = u.copy(), v.copy()
    decay = F(1.0)
    color = sample(a, u, v)
    for _ in range(16):
        cu = (cu - du).astype(F)
        cv = (cv - dv).astype(F)
        col = sample(a, cu, cv)
        color = (color + col * decay * F(weight)).astype(F)
        decay = F(decay * F(0.95))
    return (color * F(0.7)).astype(F)


def blur3(src: np.ndarray, horizontal: bool) -> np.ndarray:
    H, W = src.shape[:2]
    ox = F(F(1.0 / W) * F(1.33333333)) if horizontal else F(0)
    oy = F(0) if horizontal else F(F(1.0 / H) * F(1.33333333))
    u, v = _grid(W, H)
    s = sample(src, u, v) * F(0.29411764)
    s = s + sample(src, u - ox, v - oy) * F(0.35294117)
    s = s + sample(src, u + ox, v + oy) * F(0.35294117)
    return s.astype(F)


def apply_sunrays(dye, sw, sh, weight):
    mask = sunrays_mask(dye)
    s = sunrays_march(mask, sw, sh, weight)
    s = blur3(blur3(s, True), False)                            # blur(sunrays, sunraysTemp, 1)
    return s, mask


# ---- display: script.js:1296-1344, shader 549-612 --------------------------------------------------------------------
def _length3(c):
    return np.sqrt((c[..., 0] * c[..., 0] + c[..., 1] * c[..., 1] + c[..., 2] * c[..., 2]).astype(F)).astype(F)


def linear_to_gamma(c):
    c = np.maximum(c, F(0))
    return np.maximum(F(1.055) * np.power(c, F(0.416666667)).astype(F) - F(0.055), F(0)).astype(F)


def display(dye, w, h, shading, bloom=None, sunrays=None, dither=None, transparent=False, back=(0.0, 0.0, 0.0)):
    """drawColor + drawDisplay into a w x h float target (target != null branch of render())"""
    u, v = _grid(w, h)
    c = sample(dye, u, v)[..., :3]
    if shading:
        tx, ty = F(1.0 / w), F(1.0 / h)
        lc, rc = sample(dye, u - tx, v)[..., :3], sample(dye, u + tx, v)[..., :3]
        tc, bc = sample(dye, u, v + ty)[..., :3], sample(dye, u, v - ty)[..., :3]
        dx = (_length3(rc) - _length3(lc)).astype(F)
        dy = (_length3(tc) - _length3(bc)).astype(F)
        lz = F(np.sqrt(F(tx * tx + ty * ty)))
        nz = (lz / np.sqrt((dx * dx + dy * dy + lz * lz).astype(F))).astype(F)
        diffuse = np.clip(nz + F(0.7), F(0.7), F(1.0)).astype(F)
        c = (c * diffuse[..., None]).astype(F)
    bl = None
    if bloom is not None:
        bl = sample(bloom, u, v)[..., :3]
    if sunrays is not None:
        s = sample(sunrays, u, v)
        c = (c * s[..., None]).astype(F)
        if bl is not None:
            bl = (bl * s[..., None]).astype(F)
    if bl is not None:
        if dither is None:
            dither = np.ones((1, 1), F)                          # the 1 x 1 white placeholder, script.js:1135
        dh_, dw_ = dither.shape
        sx, sy = F(w / dw_), F(h / dh_)                          # getTextureScale, script.js:1626-1631
        noise = sample(dither, (u * sx).astype(F), (v * sy).astype(F), repeat=True)
        noise = (noise * F(2.0) - F(1.0)).astype(F)
        bl = (bl + (noise / F(255.0))[..., None]).astype(F)
        c = (c + linear_to_gamma(bl)).astype(F)
    a = np.maximum(c[..., 0], np.maximum(c[..., 1], c[..., 2])).astype(F)
    out = np.empty((h, w, 4), F)
    if transparent:                                              # blending off, nothing drawn underneath
        out[..., :3], out[..., 3] = c, a
    else:                                                        # back colour, then ONE / ONE_MINUS_SRC_ALPHA
        bk = np.array([back[0] / 255.0, back[1] / 255.0, back[2] / 255.0], dtype=np.float64).astype(F)
        out[..., :3] = (c + bk[None, None, :] * (F(1.0) - a)[..., None]).astype(F)
        out[..., 3] = (a + F(1.0) * (F(1.0) - a)).astype(F)
    return out


def normalize_texture(frame: np.ndarray) -> np.ndarray:
    """normalizeTexture, script.js:309-323: clamp01 * 255 truncated into a Uint8Array, rows flipped (top row first)"""
    x = np.clip(frame.astype(np.float64), 0.0, 1.0) * 255.0
    return np.floor(x).astype(np.uint8)[::-1].copy()


def get_resolution(resolution, canvas_w, canvas_h):
    aspect = canvas_w / canvas_h
    if aspect < 1:
        aspect = 1.0 / aspect
    lo, hi = int(np.floor(resolution + 0.5)), int(np.floor(resolution * aspect + 0.5))
    return (hi, lo) if canvas_w > canvas_h else (lo, hi)


DISPLAY_DEFAULTS = {"CAPTURE_RESOLUTION": 512, "SHADING": True, "BACK_COLOR": {"r": 0, "g": 0, "b": 0}, "TRANSPARENT": False,
                    "BLOOM": True, "BLOOM_ITERATIONS": 8, "BLOOM_RESOLUTION": 256, "BLOOM_INTENSITY": 0.8, "BLOOM_THRESHOLD": 0.6,
                    "BLOOM_SOFT_KNEE": 0.7, "SUNRAYS": True, "SUNRAYS_RESOLUTION": 196, "SUNRAYS_WEIGHT": 1.0}


def dither_pattern(w: int, h: int, seed: int) -> np.ndarray:
    """the seeded R8 pattern the live harness uploads in place of the reference's blue-noise PNG"""
    s = seed & 0xFFFFFFFF
    out = np.empty(w * h, np.uint8)

    def imul(a, b):
        return ((a & 0xFFFFFFFF) * (b & 0xFFFFFFFF)) & 0xFFFFFFFF
    for q in range(w * h):
        s = (s + 0x6D2B79F5) & 0xFFFFFFFF
        t = imul(s ^ (s >> 15), 1 | s)
        t = ((t + imul(t ^ (t >> 7), 61 | t)) & 0xFFFFFFFF) ^ t
        out[q] = ((t ^ (t >> 14)) & 0xFFFFFFFF) >> 24
    return (out.reshape(h, w).astype(F) / F(255.0)).astype(F)


def capture(dye: np.ndarray, canvas_wh, config: dict, dither=None):
    """captureScreenshot() up to the PNG: returns dict(frame, frame8, bloom, sunrays, mask)"""
    cfg = dict(DISPLAY_DEFAULTS)
    cfg.update(config)
    cw, ch = canvas_wh
    w, h = get_resolution(cfg["CAPTURE_RESOLUTION"], cw, ch)
    out = {}
    bloom = sun = None
    if cfg["BLOOM"]:
        bw, bh = get_resolution(cfg["BLOOM_RESOLUTION"], cw, ch)
        bloom = apply_bloom(dye, bw, bh, cfg["BLOOM_ITERATIONS"], cfg["BLOOM_INTENSITY"], cfg["BLOOM_THRESHOLD"], cfg["BLOOM_SOFT_KNEE"])
        out["bloom"] = bloom
    if cfg["SUNRAYS"]:
        sw, sh = get_resolution(cfg["SUNRAYS_RESOLUTION"], cw, ch)
        sun, mask = apply_sunrays(dye, sw, sh, cfg["SUNRAYS_WEIGHT"])
        out["sunrays"], out["mask"] = sun, mask
    bc = cfg["BACK_COLOR"]
    out["frame"] = display(dye, w, h, cfg["SHADING"], bloom, sun, dither, cfg["TRANSPARENT"], (bc["r"], bc["g"], bc["b"]))
    out["frame8"] = normalize_texture(out["frame"])
    return out
