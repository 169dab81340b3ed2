"""TEST INFRASTRUCTURE — the simulation passes AS THE REFERENCE'S RASTERISER EVALUATES THEM (numpy, fp32, small grids).

The restatement in fluid_oracle.c (and the HIP kernels) read the shader text: a fragment at texel (i, j) sees vUv = ((i + .5) / W,
(j + .5) / H) and its neighbours exactly.  The reference does not quite: its fragment shaders receive the five varyings of
baseVertexShader (script.js:440-459) INTERPOLATED by the rasteriser it runs on, and sample every texture through a sampler — LINEAR for
velocity and dye (script.js:1045-1077).  At power-of-two grid sizes both views coincide bit for bit; at other sizes the interpolated
coordinates are an ulp or two off the texel centres and every LINEAR fetch leaks ~W 2^-22 of a neighbouring texel ("texcoord jitter").

This module restates that rasteriser arithmetic — SwiftShader as bundled with the Chromium 88 the live harness runs (oracle/live/):
  * plane-equation setup: for a varying whose vertex values are Vm at clip coordinate -1 and Vp at +1 along an axis of N pixels (the other
    axis having M):  r = 1 / (N M),  a = M r,  A = Vp a - Vm a,  C = Vm   — every step a separately rounded fp32 operation;
  * per-pixel evaluation: value(i) = A (i + .5) + C;
  * the sampler: LINEAR = the bilinear filter of fluid_oracle.c `bil` at the interpolated coordinate, NEAREST = texel floor(u N), both
    CLAMP_TO_EDGE.
The formulas were identified from the reference's own output: oracle/live/oracle_plotly.js `probeCoords` renders the five varyings of the
reference's baseVertexShader into a float target; tests/test_raster_mode.py holds `varyings()` to committed dumps at nine grid sizes and
holds the passes below to EVERY non-power-of-two golden fixture with array_equal.  With it the only difference between the live reference
and the HIP path — the tolerances of tests/tolerances.py — is accounted for, bit for bit, by one mechanism outside the shader source.
Never used by the product."""
import numpy as np

f32 = np.float32


def varying(N, M, Vm, Vp):
    """the interpolated values along an axis of N pixels (the other axis has M) of a varying that is Vm at -1 and Vp at +1"""
    r = f32(1) / (f32(N) * f32(M))
    a = f32(M) * r
    A = f32(Vp) * a - f32(Vm) * a
    x = np.arange(N, dtype=f32) + f32(0.5)
    return (A * x).astype(f32) + f32(Vm)


class Varyings:
    """vUv, vL, vR, vT, vB of baseVertexShader for a W x H target and a texelSize uniform (script.js:440-459)"""

    def __init__(self, W, H, tsx, tsy):
        tsx, tsy = f32(tsx), f32(tsy)
        z, o = f32(0), f32(1)
        self.W, self.H = W, H
        self.ux = varying(W, H, z, o)[None, :]            # vUv.x (also vT.x, vB.x)
        self.uy = varying(H, W, z, o)[:, None]            # vUv.y (also vL.y, vR.y)
        self.lx = varying(W, H, z - tsx, o - tsx)[None, :]
        self.rx = varying(W, H, z + tsx, o + tsx)[None, :]
        self.ty = varying(H, W, z + tsy, o + tsy)[:, None]
        self.by = varying(H, W, z - tsy, o - tsy)[:, None]

    def full(self, a):
        return np.broadcast_to(a, (self.H, self.W))


def _field(F):
    F = np.asarray(F, f32)
    return F[..., None] if F.ndim == 2 else F


def linear(F, u, v):
    """texture2D on a LINEAR, CLAMP_TO_EDGE texture (fluid_oracle.c `bil`)"""
    F = _field(F)
    H, W = F.shape[:2]
    u, v = np.broadcast_arrays(np.asarray(u, f32), np.asarray(v, f32))
    x = u * f32(W) - f32(0.5)
    y = v * f32(H) - f32(0.5)
    fi, fj = np.floor(x), np.floor(y)
    fx, fy = (x - fi)[..., None], (y - fj)[..., None]
    i0, j0 = fi.astype(np.int64), fj.astype(np.int64)
    ia, ib = np.clip(i0, 0, W - 1), np.clip(i0 + 1, 0, W - 1)
    ja, jb = np.clip(j0, 0, H - 1), np.clip(j0 + 1, 0, H - 1)
    a, b, c, d = F[ja, ia], F[ja, ib], F[jb, ia], F[jb, ib]
    ab = a + (b - a) * fx
    cd = c + (d - c) * fx
    return ab + (cd - ab) * fy


def nearest(F, u, v):
    """texture2D on a NEAREST, CLAMP_TO_EDGE texture"""
    F = _field(F)
    H, W = F.shape[:2]
    u, v = np.broadcast_arrays(np.asarray(u, f32), np.asarray(v, f32))
    i = np.clip(np.floor(u * f32(W)).astype(np.int64), 0, W - 1)
    j = np.clip(np.floor(v * f32(H)).astype(np.int64), 0, H - 1)
    return F[j, i]


def _sim(vel):
    H, W = vel.shape[:2]
    return Varyings(W, H, f32(1.0 / W), f32(1.0 / H))   # velocity.texelSizeX / Y: a JS double handed to uniform2f (script.js:1061-1062)


def curl(vel):   # script.js:814-833
    V = _sim(vel)
    L = linear(vel, V.lx, V.uy)[..., 1]
    R = linear(vel, V.rx, V.uy)[..., 1]
    T = linear(vel, V.ux, V.ty)[..., 0]
    B = linear(vel, V.ux, V.by)[..., 0]
    return f32(0.5) * (R - L - T + B)


def vorticity(vel, crl, curl_strength, dt):   # script.js:835-866
    V = _sim(vel)
    L = nearest(crl, V.lx, V.uy)[..., 0]
    R = nearest(crl, V.rx, V.uy)[..., 0]
    T = nearest(crl, V.ux, V.ty)[..., 0]
    B = nearest(crl, V.ux, V.by)[..., 0]
    C = nearest(crl, V.ux, V.uy)[..., 0]
    fx = f32(0.5) * (np.abs(T) - np.abs(B))
    fy = f32(0.5) * (np.abs(R) - np.abs(L))
    ln = np.sqrt(fx * fx + fy * fy) + f32(0.0001)
    fx, fy = fx / ln, fy / ln
    s = f32(curl_strength) * C
    fx, fy = fx * s, (fy * s) * f32(-1.0)
    v = linear(vel, V.ux, V.uy)
    vx = v[..., 0] + fx * f32(dt)
    vy = v[..., 1] + fy * f32(dt)
    lim = f32(1000.0)
    return np.stack([np.minimum(np.maximum(vx, -lim), lim), np.minimum(np.maximum(vy, -lim), lim)], -1).astype(f32)


def divergence(vel):   # script.js:786-812
    V = _sim(vel)
    L = linear(vel, V.lx, V.uy)[..., 0]
    R = linear(vel, V.rx, V.uy)[..., 0]
    T = linear(vel, V.ux, V.ty)[..., 1]
    B = linear(vel, V.ux, V.by)[..., 1]
    C = linear(vel, V.ux, V.uy)
    L = np.where(V.full(V.lx) < 0, -C[..., 0], L)
    R = np.where(V.full(V.rx) > 1, -C[..., 0], R)
    T = np.where(V.full(V.ty) > 1, -C[..., 1], T)
    B = np.where(V.full(V.by) < 0, -C[..., 1], B)
    return f32(0.5) * (R - L + T - B)


def clear(p, value):   # script.js:508-519
    H, W = p.shape
    V = Varyings(W, H, f32(1.0 / W), f32(1.0 / H))
    return f32(value) * nearest(p, V.ux, V.uy)[..., 0]


def jacobi(p, div):   # script.js:868-890
    H, W = p.shape
    V = Varyings(W, H, f32(1.0 / W), f32(1.0 / H))
    L = nearest(p, V.lx, V.uy)[..., 0]
    R = nearest(p, V.rx, V.uy)[..., 0]
    T = nearest(p, V.ux, V.ty)[..., 0]
    B = nearest(p, V.ux, V.by)[..., 0]
    d = nearest(div, V.ux, V.uy)[..., 0]
    return (L + R + B + T - d) * f32(0.25)


def gradsub(p, vel):   # script.js:892-913
    V = _sim(vel)
    L = nearest(p, V.lx, V.uy)[..., 0]
    R = nearest(p, V.rx, V.uy)[..., 0]
    T = nearest(p, V.ux, V.ty)[..., 0]
    B = nearest(p, V.ux, V.by)[..., 0]
    v = linear(vel, V.ux, V.uy)
    return np.stack([v[..., 0] - (R - L), v[..., 1] - (T - B)], -1).astype(f32)


def advect(vel, src, dt, dissipation):   # script.js:746-784; the texel size of the back-trace is the SIM grid's (1276)
    src_ = _field(src)
    H, W = src_.shape[:2]
    vh, vw = vel.shape[:2]
    tsx, tsy = f32(1.0 / vw), f32(1.0 / vh)
    V = Varyings(W, H, tsx, tsy)
    v = linear(vel, V.ux, V.uy)
    cu = V.ux - f32(dt) * v[..., 0] * tsx
    cv = V.uy - f32(dt) * v[..., 1] * tsy
    decay = f32(1.0) + f32(dissipation) * f32(dt)
    out = linear(src_, cu, cv) / decay
    return out.astype(f32) if np.asarray(src).ndim == 3 else out[..., 0].astype(f32)


def exp_reference(x):
    """fluid_oracle.c fo_exp_reference (SwiftShader's exponential2), vectorised"""
    x = np.asarray(x, f32)
    x0 = f32(1.44269504) * x
    x0 = np.minimum(x0, np.array(0x43010000, np.uint32).view(f32))
    x0 = np.maximum(x0, np.array(0xC2FDFFFF, np.uint32).view(f32))
    i = np.rint(x0 - f32(0.5)).astype(np.int32)          # round to nearest even
    ii = ((i + 127).astype(np.uint32) << np.uint32(23)).view(f32)
    f = x0 - i.astype(f32)
    ff = np.full_like(f, np.array(0x3AF61905, np.uint32).view(f32))
    for c in (0x3C134806, 0x3D64AA23, 0x3E75EAD4, 0x3F31727B):
        ff = ff * f + np.array(c, np.uint32).view(f32)
    ff = ff * f + f32(1.0)
    return ii * ff


def splat(base, x, y, aspect, radius, color):   # script.js:726-744
    base = np.asarray(base, f32)
    H, W, nc = base.shape
    V = Varyings(W, H, f32(1.0 / W), f32(1.0 / H))
    px = (V.ux - f32(x)) * f32(aspect)
    py = V.uy - f32(y)
    g = exp_reference(-(px * px + py * py) / f32(radius))
    b = linear(base, V.ux, V.uy)
    out = np.empty_like(base)
    for k in range(min(nc, 3)):
        out[..., k] = b[..., k] + g * f32(color[k])
    if nc == 4:
        out[..., 3] = 1.0
    return out


def resample(src, newW, newH):   # copyProgram through resizeFBO, script.js:496-506, 1108-1114
    V = Varyings(newW, newH, f32(1.0 / newW), f32(1.0 / newH))
    out = linear(src, V.ux, V.uy)
    return out.astype(f32) if np.asarray(src).ndim == 3 else out[..., 0].astype(f32)


def _api():
    """the part of oracle.py's interface tests/scenario.py OracleAdapter drives, on the rasteriser's arithmetic"""
    from . import oracle as O

    class RasterSim(O.RefSim):
        def init_framebuffers(self):
            keep = O.resample
            O.resample = resample            # RefSim resizes through the module-level function
            try:
                super().init_framebuffers()
            finally:
                O.resample = keep

        def splat(self, x, y, dx, dy, color):
            aspect = self.canvas[0] / self.canvas[1]
            radius = self.config["SPLAT_RADIUS"] / 100.0
            if aspect > 1:
                radius *= aspect
            a, r = O.f32(aspect), O.f32(radius)
            self.vel[0] = O.stored(splat(self.vel[0], O.f32(x), O.f32(y), a, r, (O.f32(dx), O.f32(dy), 0.0)), self.storage)
            self.dye[0] = O.stored(splat(self.dye[0], O.f32(x), O.f32(y), a, r, tuple(O.f32(c) for c in color)), self.storage)

        def step(self, dt=0.016666, n=1):   # script.js:1231-1294
            P = self.params()
            st = lambda a: O.stored(np.ascontiguousarray(a, f32), self.storage)  # noqa: E731
            for _ in range(n):
                self.curl = st(curl(self.vel[0]))
                self.vel[0] = st(vorticity(self.vel[0], self.curl, P.curl, dt))
                self.div = st(divergence(self.vel[0]))
                self.prs[0] = st(clear(self.prs[0], P.pressure))
                for _ in range(P.iterations):
                    self.prs[0] = st(jacobi(self.prs[0], self.div))
                self.vel[0] = st(gradsub(self.prs[0], self.vel[0]))
                self.vel[0] = st(advect(self.vel[0], self.vel[0], dt, P.velocity_dissipation))
                self.dye[0] = st(advect(self.vel[0], self.dye[0], dt, P.density_dissipation))

    import types
    ns = types.SimpleNamespace(RefSim=RasterSim, stored=O.stored, f32=O.f32, curl=curl, vorticity=vorticity, divergence=divergence,
                               clear=clear, jacobi=jacobi, gradsub=gradsub, advect=advect, splat=splat, resample=resample)
    return ns
