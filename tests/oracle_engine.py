"""TEST INFRASTRUCTURE — a stripe compute engine backed by the CPU oracle, with the same interface as
fluid_hip.stripes.HipStripeEngine.  Injected into StripeSim so the decomposition / ghost-row host logic
can be exercised on CPU (threads + LocalComm, or gloo world_size 2)."""
import numpy as np
import torch

from oracle import oracle as O


class _Info:
    def __init__(self, width, height, channels, row0, rows, halo):
        self.width, self.height, self.channels, self.row0, self.rows, self.halo = width, height, channels, row0, rows, halo


class OracleStripeEngine:
    def __init__(self, sim_wh, dye_wh, part, parts, halo, schedule, device):
        (self.W, self.H), (self.DW, self.DH) = sim_wh, dye_wh
        self.parts, self.part = parts, part
        self.halo = halo if parts > 1 else 0
        self.rows, self.drows = self.H // parts, self.DH // parts
        self.row0, self.drow0 = part * self.rows, part * self.drows
        self.dhalo = (self.halo * self.DH + self.H - 1) // self.H if parts > 1 else 0
        self.g0, self.dg0 = self.row0 - self.halo, self.drow0 - self.dhalo
        n, dn = self.rows + 2 * self.halo, self.drows + 2 * self.dhalo
        self.vel = np.zeros((n, self.W, 2), np.float32)
        self.prs = np.zeros((n, self.W), np.float32)
        self.div = np.zeros((n, self.W), np.float32)
        self.crl = np.zeros((n, self.W), np.float32)
        self.dye = np.zeros((dn, self.DW, 4), np.float32)
        self.dye[..., 3] = 1.0
        self.misses = 0

    def close(self):
        pass

    def stream_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    def _arr(self, name):
        return {"velocity": self.vel, "pressure": self.prs, "divergence": self.div, "curl": self.crl, "dye": self.dye}[name]

    def _set(self, name, a):
        setattr(self, {"velocity": "vel", "pressure": "prs", "divergence": "div", "curl": "crl", "dye": "dye"}[name], a)

    def info(self, name):
        if name == "dye":
            return _Info(self.DW, self.DH, 4, self.drow0, self.drows, self.dhalo)
        return _Info(self.W, self.H, 2 if name == "velocity" else 1, self.row0, self.rows, self.halo)

    # local array row range of [row0 - ext, row0 + rows + ext) clipped to the domain
    def _range(self, ext, dye=False):
        row0, rows, g0, H = (self.drow0, self.drows, self.dg0, self.DH) if dye else (self.row0, self.rows, self.g0, self.H)
        ga, gb = max(row0 - ext, 0), min(row0 + rows + ext, H)
        return ga - g0, gb - g0

    def view(self, name):
        """torch tensor aliasing the numpy window array (zero copy), [array rows, W, channels]"""
        a = self._arr(name)
        t = torch.from_numpy(a)
        return t if a.ndim == 3 else t.unsqueeze(-1)

    def curl(self, ext):
        ra, rb = self._range(ext)
        out = O.curl(self.vel, H=self.H, g0=self.g0, ra=ra, rb=rb)
        self.crl[ra:rb] = out[ra:rb]

    def vorticity(self, curl, dt, ext):
        ra, rb = self._range(ext)
        self.vel = O.vorticity(self.vel, self.crl, O.f32(curl), O.f32(dt), H=self.H, g0=self.g0, ra=ra, rb=rb)

    def divergence(self, ext):
        ra, rb = self._range(ext)
        out = O.divergence(self.vel, H=self.H, g0=self.g0, ra=ra, rb=rb)
        self.div[ra:rb] = out[ra:rb]

    def curl_vorticity_divergence(self, curl, dt, ext):
        self.curl(ext + 2)
        self.vorticity(curl, dt, ext + 1)
        self.divergence(ext)

    def clear(self, value, ext):
        ra, rb = self._range(ext)
        self.prs = O.clear(self.prs, O.f32(value), ra=ra, rb=rb)

    def jacobi(self, iters, ext_out):
        for k in range(iters):
            ra, rb = self._range(ext_out + iters - 1 - k)
            self.prs = O.jacobi(self.prs, self.div, H=self.H, g0=self.g0, ra=ra, rb=rb)

    def clear_jacobi(self, value, iters, ext_out):
        self.clear(value, ext_out + iters)
        self.jacobi(iters, ext_out)

    def gradsub(self, ext):
        ra, rb = self._range(ext)
        self.vel = O.gradsub(self.prs, self.vel, H=self.H, g0=self.g0, ra=ra, rb=rb)

    def advect_velocity(self, dt, diss, ext):
        ra, rb = self._range(ext)
        self.vel, m = O.advect(self.vel, self.vel, O.f32(dt), O.f32(diss), vH=self.H, vg0=self.g0, sH=self.H, sg0=self.g0,
                               ra=ra, rb=rb, return_misses=True)
        self.misses += m

    def advect_dye(self, dt, diss):
        ra, rb = self._range(0, dye=True)
        self.dye, m = O.advect(self.vel, self.dye, O.f32(dt), O.f32(diss), vH=self.H, vg0=self.g0, sH=self.DH, sg0=self.dg0,
                               ra=ra, rb=rb, return_misses=True)
        self.misses += m

    def advect(self, dt, vdiss, ddiss):
        same = (self.W, self.H) == (self.DW, self.DH)
        self.advect_velocity(dt, vdiss, 1 if (self.parts > 1 and not same) else 0)
        self.advect_dye(dt, ddiss)

    def splat(self, x, y, dx, dy, r, g, b, aspect, radius):
        f = O.f32
        ra, rb = self._range(self.halo)
        self.vel = O.splat(self.vel, f(x), f(y), f(aspect), f(radius), (f(dx), f(dy), 0.0), H=self.H, g0=self.g0, ra=ra, rb=rb)
        ra, rb = self._range(self.dhalo, dye=True)
        self.dye = O.splat(self.dye, f(x), f(y), f(aspect), f(radius), (f(r), f(g), f(b)), H=self.DH, g0=self.dg0, ra=ra, rb=rb)

    def read(self, name):
        fi, a = self.info(name), self._arr(name)
        return a[fi.halo:fi.halo + fi.rows].copy()

    def write(self, name, arr):
        fi, a = self.info(name), self._arr(name)
        a[fi.halo:fi.halo + fi.rows] = arr

    def sync(self):
        pass

    def check_halo(self):
        if self.misses:
            m, self.misses = self.misses, 0
            raise RuntimeError("%d advection taps fell outside the stripe's ghost rows" % m)
