"""Row-stripe domain decomposition of the reference's step() across GPUs (one process per GPU).

The global grid is cut into `world` equal row stripes.  Each rank owns one stripe plus `halo` ghost
rows on each side and runs the SAME kernels as the single-GPU path on its window (the C ABI's
`part/parts/halo`).  Between passes, ghost rows are refreshed from the neighbouring ranks with
point-to-point send/recv — torch.distributed, i.e. RCCL over xGMI on MI355X, gloo on CPU for the
tests.  There is no global collective on the data path.

Communication-avoiding schedule (MI355X-first: xGMI is point-to-point and a neighbour message is
tiny, so the cost is per-exchange latency, not bytes): instead of one single-row exchange before
each of the 7 + ITERS passes, a rank recomputes a few ghost rows redundantly and exchanges `halo`
rows at a time:

    exchange { velocity (halo rows), pressure (D+e rows) }         one batched send/recv
    curl -> vorticity -> divergence, fused (ext halo-3)           no exchange
    clear (ext D+e: the ghost rows hold the neighbour's pre-clear pressure)
    D Jacobi iterations; if iterations remain: exchange pressure, next block   (D = min(remaining, halo-3))
    gradient subtract (ext 0; the last Jacobi block left e = 1 valid ghost row of pressure)
    exchange { velocity (halo rows), dye (dye-halo rows) }       one batched send/recv
    advect velocity + dye (one kernel when the dye grid is the sim grid)

With halo = 32 and 50 iterations that is 3 batched exchanges per step (2 with halo >= 54) instead of 57
single-row ones.  Ghost rows are sent and received IN PLACE: the exchange operates on torch views of the field
arrays (rows are contiguous), so there is no pack/unpack copy and no staging allocation in the step loop.
Every recomputed ghost
row is the same arithmetic on the same inputs as the owner's, so the decomposed result is BITWISE
equal to the single-domain result (tests/test_stripes_*.py).  `halo` must cover the advection
back-trace (dt * max|v_y| + 2 rows); kernels count taps that leave the window and `check_halo()`
raises instead of returning a wrong field.

The compute engine is injectable (`engine_factory`) so the host logic here can be exercised on CPU
by the tests with the oracle as the engine; the default engine is the HIP C ABI and nothing else.

Two drivers run this schedule on the HIP engine:
  * NATIVE (default under torch.distributed's nccl backend): libfluid_hip.so executes the whole plan itself —
    `fluid_step_n` on the stripe context, ncclSend/ncclRecv issued from C++ on the context stream (csrc/
    fluid_stripes.cpp, `fluid_stripe_plan` is the same schedule as `StripeSim.step` below and is held to it by
    tests/test_stripes_cpu.py).  torch.distributed only carries the 128-byte ncclUniqueId at start-up.
  * HOSTED (`native=False`, and always for the CPU/oracle engine): this module calls the passes one by one and
    exchanges through `comm.exchange` (torch.distributed batch_isend_irecv, or LocalComm mailboxes).
"""
from __future__ import annotations

import ctypes as C
import queue
import random as _random
from typing import Callable, List, Optional

import numpy as np

from . import _abi
from ._abi import FIELD_IDS
from .sim import Canvas, DEFAULT_CONFIG, HSVtoRGB, getResolution

VELOCITY, PRESSURE, DYE = "velocity", "pressure", "dye"


# ---------------------------------------------------------------------------------------------------
class HipStripeEngine:
    """One stripe context of libfluid_hip.so; ghost rows are staged through torch device tensors and all
    work is enqueued on torch's current stream so RCCL send/recv order correctly against the kernels."""

    def __init__(self, sim_wh, dye_wh, part, parts, halo, schedule, device, part_x=0, parts_x=1, storage="f32"):
        import torch
        self.torch = torch
        self.lib = _abi.lib()
        self.device = device
        d = _abi.Desc(sim_wh[0], sim_wh[1], dye_wh[0], dye_wh[1], device, part, parts, halo, schedule, part_x, parts_x, _abi.STORAGE[storage])
        ctx = C.c_void_p()
        rc = self.lib.fluid_create(C.byref(d), C.byref(ctx))
        if rc != _abi.FLUID_OK:
            _abi.check(None, rc)
        self.ctx = ctx
        self._views = {}
        self._info = {}
        # a dedicated non-blocking torch stream: kernels, ghost-row views and the RCCL send/recv issued under
        # `stream_ctx()` are all ordered on it (launches on the legacy null stream cost far more host time)
        self.stream = torch.cuda.Stream(device=device)
        self._ck(self.lib.fluid_set_stream(self.ctx, C.c_void_p(self.stream.cuda_stream), 1))

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def _ck(self, rc):
        _abi.check(self.ctx, rc)

    def close(self):
        if self.ctx is not None:
            self.lib.fluid_destroy(self.ctx)
            self.ctx = None

    def info(self, name):
        fi = self._info.get(name)
        if fi is None:
            fi = _abi.FieldInfo()
            self._ck(self.lib.fluid_field_info_get(self.ctx, FIELD_IDS[name], C.byref(fi)))
            self._info[name] = fi
        return fi

    def view(self, name):
        """torch tensor [array rows (ghost rows included), W, channels] aliasing the field's CURRENT read buffer
        (zero copy, through __cuda_array_interface__; both ping-pong buffers are cached by address)"""
        ptr = C.c_void_p()
        self._ck(self.lib.fluid_field_device_ptr(self.ctx, FIELD_IDS[name], C.byref(ptr)))
        key = (name, ptr.value)
        t = self._views.get(key)
        if t is None:
            fi = self.info(name)
            shape = (fi.rows + 2 * fi.halo, fi.pitch, fi.channels)   # array rows x pitch (>= width: rows stay 16-byte aligned)

            class _DeviceArray:  # minimal CUDA-array-interface carrier
                __cuda_array_interface__ = {"shape": shape, "typestr": "<f%d" % fi.bytes_per_channel, "data": (ptr.value, False), "version": 2}

            t = self.torch.as_tensor(_DeviceArray(), device="cuda:%d" % self.device)
            self._views[key] = t
        return t

    def curl(self, ext): self._ck(self.lib.fluid_pass_curl(self.ctx, ext))
    def vorticity(self, curl, dt, ext): self._ck(self.lib.fluid_pass_vorticity(self.ctx, curl, dt, ext))
    def divergence(self, ext): self._ck(self.lib.fluid_pass_divergence(self.ctx, ext))
    def curl_vorticity_divergence(self, curl, dt, ext): self._ck(self.lib.fluid_pass_curl_vorticity_divergence(self.ctx, curl, dt, ext))
    def clear(self, value, ext): self._ck(self.lib.fluid_pass_clear(self.ctx, value, ext))
    def jacobi(self, iters, ext_out): self._ck(self.lib.fluid_pass_jacobi(self.ctx, iters, ext_out))
    def clear_jacobi(self, value, iters, ext_out): self._ck(self.lib.fluid_pass_clear_jacobi(self.ctx, value, iters, ext_out))
    def gradsub(self, ext): self._ck(self.lib.fluid_pass_gradsub(self.ctx, ext))
    def advect_velocity(self, dt, diss, ext): self._ck(self.lib.fluid_pass_advect_velocity(self.ctx, dt, diss, ext))
    def advect_dye(self, dt, diss): self._ck(self.lib.fluid_pass_advect_dye(self.ctx, dt, diss))
    def advect(self, dt, vdiss, ddiss): self._ck(self.lib.fluid_pass_advect(self.ctx, dt, vdiss, ddiss))

    def splat(self, x, y, dx, dy, r, g, b, aspect, radius):
        self._ck(self.lib.fluid_splat(self.ctx, x, y, dx, dy, r, g, b, aspect, radius))

    def read(self, name):
        fi = self.info(name)   # the owned rows x owned columns (all columns unless the context is a 2-D tile)
        shape = (fi.rows, fi.cols) if fi.channels == 1 else (fi.rows, fi.cols, fi.channels)
        out = np.empty(shape, np.float32)
        self._ck(self.lib.fluid_read_field(self.ctx, FIELD_IDS[name], out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def write(self, name, arr):
        a = np.ascontiguousarray(arr, np.float32)
        self._ck(self.lib.fluid_write_field(self.ctx, FIELD_IDS[name], a.ctypes.data_as(C.c_void_p), a.nbytes))

    def sync(self): self._ck(self.lib.fluid_sync(self.ctx))
    def check_halo(self): self._ck(self.lib.fluid_halo_check(self.ctx))

    # -- native driver: the library runs the plan and the RCCL exchanges itself ---------------------------------
    def set_step_marks(self, capacity):
        self._ck(self.lib.fluid_set_step_marks(self.ctx, int(capacity)))

    def step_marks(self):
        n = C.c_int(0)
        self._ck(self.lib.fluid_get_step_marks(self.ctx, None, 0, C.byref(n)))
        buf = (C.c_float * max(n.value, 1))()
        self._ck(self.lib.fluid_get_step_marks(self.ctx, buf, n.value, C.byref(n)))
        return [float(buf[k]) for k in range(n.value)]

    def use_own_stream(self):
        """back to the context's own HIP stream (the native driver does not involve torch streams)"""
        self._ck(self.lib.fluid_set_stream(self.ctx, None, 0))
        self._views.clear()

    def comm_init(self, id_bytes: bytes):
        cid = _abi.CommId()
        C.memmove(C.byref(cid), id_bytes, 128)
        self._ck(self.lib.fluid_comm_init(self.ctx, C.byref(cid)))

    def comm_selftest(self, nfloats=1 << 16):
        self._ck(self.lib.fluid_comm_selftest(self.ctx, nfloats))

    def step_n(self, n, dt, config):
        p = _abi.Params(float(config["CURL"]), float(config["PRESSURE"]), int(config["PRESSURE_ITERATIONS"]),
                        float(config["VELOCITY_DISSIPATION"]), float(config["DENSITY_DISSIPATION"]))
        self._ck(self.lib.fluid_step_n(self.ctx, int(n), float(dt), C.byref(p)))

    def exchange_count(self):
        return int(self.lib.fluid_exchange_count(self.ctx))

    def schedule_info(self, n_steps, dt, config):
        """fluid_schedule_info_get for this context (tile shape, folds, whether the dye is / would be packed)"""
        p = _abi.Params(float(config["CURL"]), float(config["PRESSURE"]), int(config["PRESSURE_ITERATIONS"]),
                        float(config["VELOCITY_DISSIPATION"]), float(config["DENSITY_DISSIPATION"]))
        info = _abi.ScheduleInfo()
        self._ck(self.lib.fluid_schedule_info_get(self.ctx, int(n_steps), float(dt), C.byref(p), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _abi.ScheduleInfo._fields_}

    def set_reach(self, rows): self._ck(self.lib.fluid_set_reach(self.ctx, int(rows)))
    def set_overlap(self, on): self._ck(self.lib.fluid_set_overlap(self.ctx, 1 if on else 0))
    def set_link_model(self, latency_us, gbytes_per_s): self._ck(self.lib.fluid_set_link_model(self.ctx, float(latency_us), float(gbytes_per_s)))

    def calibrate_link(self, reps=20):
        """collective over the stripe / tile set (behind comm_init): measure what a neighbour message costs and make it this context's
        link model (fluid_comm_calibrate_link) -> (latency_us, GB/s)"""
        lat, bw = C.c_float(0), C.c_float(0)
        self._ck(self.lib.fluid_comm_calibrate_link(self.ctx, int(reps), C.byref(lat), C.byref(bw)))
        return float(lat.value), float(bw.value)

    def advect_exchange_rows(self):
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self.lib.fluid_advect_exchange_rows(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value


def new_comm_id() -> bytes:
    """ncclGetUniqueId through libfluid_hip (rank 0 calls this and ships the 128 bytes to the other ranks)"""
    cid = _abi.CommId()
    _abi.check(None, _abi.lib().fluid_comm_unique_id(C.byref(cid)))
    return C.string_at(C.byref(cid), 128)


# ---------------------------------------------------------------------------------------------------
class TorchDistComm:
    """neighbour exchange over torch.distributed point-to-point (backend nccl = RCCL over xGMI; gloo on CPU).
    One batch_isend_irecv per exchange, operating in place on views of the field arrays."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = str(dist.get_backend(group))

    def broadcast_bytes(self, payload: Optional[bytes]) -> bytes:
        """rank 0's bytes on every rank (start-up only: carries the ncclUniqueId of the native driver)"""
        box = [payload]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        """lists of tensors: send_lo[i] -> rank-1 (lands in its recv_hi[i]); send_hi[i] -> rank+1 (its recv_lo[i])"""
        dist, ops = self.dist, []
        if self.rank > 0:
            for snd, rcv in zip(send_lo, recv_lo):
                ops += [dist.P2POp(dist.isend, snd, self.rank - 1, self.group), dist.P2POp(dist.irecv, rcv, self.rank - 1, self.group)]
        if self.rank < self.world - 1:
            for snd, rcv in zip(send_hi, recv_hi):
                ops += [dist.P2POp(dist.isend, snd, self.rank + 1, self.group), dist.P2POp(dist.irecv, rcv, self.rank + 1, self.group)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()  # nccl: the current stream waits for the transfer; gloo: blocks the host

    def gather_rows(self, arr: np.ndarray) -> Optional[np.ndarray]:
        parts = [None] * self.world
        self.dist.all_gather_object(parts, arr, group=self.group)
        return np.concatenate(parts, axis=0)

    def gather_tiles(self, arr: np.ndarray, tiles_x: int) -> np.ndarray:
        """2-D decomposition (rank = stripe * tiles_x + tile column): the global field on every rank"""
        parts = [None] * self.world
        self.dist.all_gather_object(parts, arr, group=self.group)
        return np.concatenate([np.concatenate(parts[y * tiles_x:(y + 1) * tiles_x], axis=1) for y in range(self.world // tiles_x)], axis=0)


class LocalComm:
    """all stripes inside ONE process (one thread per stripe, all contexts on one device): the way the
    decomposition is validated bit-for-bit on a single-GPU box.  Blocking mailboxes between neighbours;
    the sender posts CLONES (taken in its own stream order), the receiver copies them into its ghost rows."""

    class Hub:
        def __init__(self, world):
            self.world = world
            self.box = {(s, d): queue.Queue() for s in range(world) for d in (s - 1, s + 1) if 0 <= d < world}

    def __init__(self, hub: "LocalComm.Hub", rank: int):
        self.hub, self.rank, self.world = hub, rank, hub.world

    def _post(self, dst, tensors):
        clones, ev = [t.clone() for t in tensors], None
        if clones and clones[0].is_cuda:  # each stripe runs on its own stream: hand the clones over with an event
            import torch
            ev = torch.cuda.Event()
            ev.record()
        self.hub.box[(self.rank, dst)].put((clones, ev))

    def _take(self, src, recvs):
        clones, ev = self.hub.box[(src, self.rank)].get(timeout=120)
        if ev is not None:
            import torch
            torch.cuda.current_stream().wait_event(ev)
        for dst, c in zip(recvs, clones):
            dst.copy_(c)

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        r = self.rank
        if r > 0:
            self._post(r - 1, send_lo)
        if r < self.world - 1:
            self._post(r + 1, send_hi)
        if r > 0:
            self._take(r - 1, recv_lo)
        if r < self.world - 1:
            self._take(r + 1, recv_hi)

    def gather_rows(self, arr):
        raise NotImplementedError("gather the per-stripe reads in the caller")


# ---------------------------------------------------------------------------------------------------
class StripeSim:
    """This rank's stripe of a FluidSim: same surface (config / splat / multipleSplats / step / read)."""

    def __init__(self, canvas=(512, 512), config: Optional[dict] = None, halo: int = 32, schedule: str = "fused",
                 random: Optional[Callable[[], float]] = None, device: int = 0, comm=None,
                 engine_factory: Optional[Callable] = None, native: Optional[bool] = None, reach: Optional[int] = None,
                 overlap: Optional[bool] = None, tiles_x: int = 1, storage: str = "f32", link_model=None):
        """tiles_x > 1: 2-D decomposition, world // tiles_x row stripes x tiles_x column tiles, rank = stripe * tiles_x +
        tile column (native driver only; the hosted schedule below is 1-D).  link_model = (latency_us, GB/s) of one neighbour
        message (fluid_set_link_model: sizes how much compute the native driver puts in front of an exchange's arrival);
        link_model = "calibrate": measure it at start-up on this set's own links (fluid_comm_calibrate_link; `self.link_model` holds
        what came out); None: the library's defaults"""
        self._link_model = link_model
        self.link_model = None   # (latency_us, GB/s, source) once known
        self.canvas = canvas if isinstance(canvas, Canvas) else Canvas(*canvas)
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.random = random or _random.random
        self.comm = comm if comm is not None else TorchDistComm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.tiles_x = int(tiles_x)
        self.storage = storage
        if self.tiles_x > 1:
            return self._init_tiles(halo, schedule, device, reach)
        sim = getResolution(self.config["SIM_RESOLUTION"], self.canvas.width, self.canvas.height)
        dye = getResolution(self.config["DYE_RESOLUTION"], self.canvas.width, self.canvas.height)
        self.sim_wh = (sim["width"], sim["height"])
        self.dye_wh = (dye["width"], dye["height"])
        if self.sim_wh[1] % self.world or self.dye_wh[1] % self.world:
            raise ValueError("grid heights %d / %d do not divide into %d stripes" % (self.sim_wh[1], self.dye_wh[1], self.world))
        self.halo = int(halo) if self.world > 1 else 0
        if self.world > 1 and self.halo < 4:
            raise ValueError("halo must be >= 4")
        sched = {"passes": _abi.SCHED_PASSES, "fused": _abi.SCHED_FUSED}[schedule]
        if engine_factory is None:
            self.engine = HipStripeEngine(self.sim_wh, self.dye_wh, self.rank, self.world, self.halo, sched, device, storage=storage)
        else:
            self.engine = engine_factory(self.sim_wh, self.dye_wh, self.rank, self.world, self.halo, sched, device)
        self.same_res = self.sim_wh == self.dye_wh
        self._hosted_exchanges = 0
        # native driver: default whenever the ranks are connected by RCCL (torch.distributed backend nccl)
        if native is None:
            native = (engine_factory is None and isinstance(self.comm, TorchDistComm) and self.comm.backend == "nccl")
        self.native = bool(native)
        if self.native:
            self._native_comm(reach, overlap)

    def _init_tiles(self, halo, schedule, device, reach):
        if self.world % self.tiles_x:
            raise ValueError("world size %d is not a multiple of tiles_x %d" % (self.world, self.tiles_x))
        sim = getResolution(self.config["SIM_RESOLUTION"], self.canvas.width, self.canvas.height)
        dye = getResolution(self.config["DYE_RESOLUTION"], self.canvas.width, self.canvas.height)
        self.sim_wh, self.dye_wh = (sim["width"], sim["height"]), (dye["width"], dye["height"])
        self.halo = int(halo)
        sched = {"passes": _abi.SCHED_PASSES, "fused": _abi.SCHED_FUSED}[schedule]
        self.engine = HipStripeEngine(self.sim_wh, self.dye_wh, self.rank // self.tiles_x, self.world // self.tiles_x, self.halo, sched,
                                      device, part_x=self.rank % self.tiles_x, parts_x=self.tiles_x, storage=self.storage)
        self.same_res = self.sim_wh == self.dye_wh
        self._hosted_exchanges = 0
        self.native = True
        self._native_comm(reach, None)

    def _native_comm(self, reach, overlap):
        self.engine.use_own_stream()
        if reach is not None:
            self.engine.set_reach(reach)
        if overlap is not None:
            self.engine.set_overlap(overlap)
        lm = getattr(self, "_link_model", None)
        if lm is not None and not isinstance(lm, str):
            self.engine.set_link_model(*lm)
            self.link_model = (float(lm[0]), float(lm[1]), "given")
        payload = None
        if self.rank == 0:   # a failure on rank 0 must reach every rank, or they would wait in the broadcast forever
            try:
                payload = new_comm_id()
            except _abi.FluidError as ex:
                payload = b"ERR:" + str(ex).encode()
        payload = self.comm.broadcast_bytes(payload)
        if payload[:4] == b"ERR:":
            raise _abi.FluidError(_abi.ERR_COMM, payload[4:].decode())
        self.engine.comm_init(payload)
        if lm == "calibrate" and self.world > 1:
            try:
                lat, bw = self.engine.calibrate_link()
                self.link_model = (lat, bw, "measured at start-up: fluid_comm_calibrate_link, 20 exchanges of 4 KB and of the step's largest message")
            except _abi.FluidError as ex:   # the probe is an optimisation: the library's constants stay, the line says why
                self.link_model = (20.0, 50.0, "library default; the start-up probe failed: %s" % str(ex)[:160])

    @property
    def exchanges(self):
        return self.engine.exchange_count() if self.native else self._hosted_exchanges

    def close(self):
        self.engine.close()

    # -- ghost-row refresh -------------------------------------------------------------------------
    def exchange(self, *items):
        """items: (field name, rows) pairs refreshed in ONE batched neighbour exchange, in place"""
        items = [(n, k) for n, k in items if k > 0]
        if self.world == 1 or not items:
            return
        e = self.engine
        send_lo, send_hi, recv_lo, recv_hi = [], [], [], []
        for name, n in items:
            t, fi = e.view(name), e.info(name)
            h, r = fi.halo, fi.rows
            if n > h or n > r:
                raise ValueError("exchange of %d rows exceeds halo %d / stripe %d" % (n, h, r))
            send_lo.append(t[h:h + n])              # my lowest owned rows  -> lower neighbour's top ghost rows
            recv_lo.append(t[h - n:h])              # my bottom ghost rows  <- lower neighbour's highest owned rows
            send_hi.append(t[h + r - n:h + r])
            recv_hi.append(t[h + r:h + r + n])
        with e.stream_ctx():
            self.comm.exchange(send_lo, send_hi, recv_lo, recv_hi)
        self._hosted_exchanges += 1

    # -- splat / multipleSplats: script.js:1441-1462, 1427-1439 (every rank evaluates its own rows) ----
    def splat(self, x, y, dx, dy, color):
        r, g, b = (color["r"], color["g"], color["b"]) if isinstance(color, dict) else color
        aspect = self.canvas.width / self.canvas.height
        radius = self.config["SPLAT_RADIUS"] / 100.0
        if aspect > 1:
            radius *= aspect
        self.engine.splat(x, y, dx, dy, r, g, b, aspect, radius)

    def multipleSplats(self, amount: int):
        issued = []
        for _ in range(int(amount)):  # same Math.random call order on every rank (same seed -> same stream)
            c = HSVtoRGB(self.random(), 1.0, 1.0)
            color = {k: v * 0.15 * 10.0 for k, v in c.items()}
            x = self.random()
            y = self.random()
            dx = 1000 * (self.random() - 0.5)
            dy = 1000 * (self.random() - 0.5)
            self.splat(x, y, dx, dy, color)
            issued.append([x, y, dx, dy, color["r"], color["g"], color["b"]])
        return issued

    # -- step(dt): script.js:1231-1294 with ghost-row exchanges between pass groups ---------------------
    def step(self, dt: float, n: int = 1):
        if self.native:
            self.engine.step_n(n, dt, self.config)   # the whole plan, exchanges included, inside libfluid_hip.so
            return
        for _ in range(n):
            self._step_hosted(dt)
        if n > 0:
            self.engine.check_halo()   # as the native driver does: a reach violation fails the call that caused it

    def _step_hosted(self, dt: float):
        c, e, H = self.config, self.engine, self.halo
        iters = int(c["PRESSURE_ITERATIONS"])
        if self.world == 1:
            e.curl_vorticity_divergence(c["CURL"], dt, 0)
            e.clear_jacobi(c["PRESSURE"], iters, 0); e.gradsub(0)
            e.advect(dt, c["VELOCITY_DISSIPATION"], c["DENSITY_DISSIPATION"])
            return
        # pressure blocks: divergence is valid H-3 rows out and iteration k of a block needs it d-k+e rows out
        # -> d <= H-3; the last block also produces e = 1 ghost row (gradient subtract reads pressure one row out)
        # as few blocks as the ghost rows allow, balanced in whole launches of ten iterations where those fit under the cap
        # (csrc/fluid_stripes.cpp build_plan: 200 iterations at H = 56 are 4 x 50, not 53 + 53 + 53 + 41), greedy otherwise
        cap, sizes = H - 3, []
        if iters > 0:
            nb, L, total = -(-iters // cap), -(-iters // 10), 0
            for k in range(nb):
                d = (L // nb + (1 if k < L % nb else 0)) * 10 if k < nb - 1 else iters - total
                sizes.append(d)
                total += d
            if not all(0 < d <= cap for d in sizes):
                sizes, remaining = [], iters
                while remaining > 0:
                    d = min(remaining, cap)
                    remaining -= d
                    sizes.append(d)
        blocks = [(d, 1 if k == len(sizes) - 1 else 0) for k, d in enumerate(sizes)]
        first = blocks[0][0] + blocks[0][1] if blocks else 1
        self.exchange((VELOCITY, H), (PRESSURE, first))
        e.curl_vorticity_divergence(c["CURL"], dt, H - 3)   # curl to H-1, vorticity to H-2, divergence to H-3 rows out
        if not blocks:
            e.clear(c["PRESSURE"], 1)
        for k, (d, ext) in enumerate(blocks):
            if k == 0:   # the exchanged ghost rows hold the neighbour's PRE-clear pressure: clear covers them too
                e.clear_jacobi(c["PRESSURE"], d, ext)
            else:
                self.exchange((PRESSURE, d + ext))
                e.jacobi(d, ext)
        e.gradsub(0)
        self.exchange((VELOCITY, H), (DYE, e.info(DYE).halo))
        e.advect(dt, c["VELOCITY_DISSIPATION"], c["DENSITY_DISSIPATION"])   # velocity then dye (one kernel when the grids match)

    def sync(self):
        self.engine.sync()

    def check_halo(self):
        self.engine.check_halo()

    # -- field access ------------------------------------------------------------------------------------
    def read_local(self, name: str) -> np.ndarray:
        return self.engine.read(name)

    def read(self, name: str) -> np.ndarray:
        """the GLOBAL field on every rank (test / checkpoint path; goes through the host)"""
        local = self.read_local(name)
        if self.world == 1:
            return local
        if self.tiles_x > 1:   # tiles of one stripe side by side first, then the stripes on top of each other
            local = self.comm.gather_tiles(local, self.tiles_x)
            return local
        return self.comm.gather_rows(local)

    def write(self, name: str, global_arr: np.ndarray):
        fi = self.engine.info(name)
        self.engine.write(name, np.ascontiguousarray(global_arr[fi.row0:fi.row0 + fi.rows, fi.col0:fi.col0 + fi.cols]))


class StripeGroup:
    """The WHOLE stripe set in one process, stepped by libfluid_hip's own plan (`fluid_group_step_n`: the same plan
    and kernels as the RCCL driver, ghost rows moved by device-to-device copies).  Validation on a single-GPU box."""

    def __init__(self, world: int, canvas=(512, 512), config: Optional[dict] = None, halo: int = 32, schedule: str = "fused",
                 random: Optional[Callable[[], float]] = None, device: int = 0, reach: Optional[int] = None,
                 overlap: Optional[bool] = None, tiles_x: int = 1, storage: str = "f32", link_model=None):
        """`world` contexts: world // tiles_x row stripes x tiles_x column tiles (tiles_x = 1: the 1-D stripe set);
        link_model = (latency_us, GB/s): fluid_set_link_model on every context"""
        if world % tiles_x:
            raise ValueError("world must be a multiple of tiles_x")
        self.tiles_x, self.tiles_y = tiles_x, world // tiles_x
        self.canvas = canvas if isinstance(canvas, Canvas) else Canvas(*canvas)
        self.config = dict(DEFAULT_CONFIG)
        if config:
            self.config.update(config)
        self.random = random or _random.random
        self.world = world
        sim = getResolution(self.config["SIM_RESOLUTION"], self.canvas.width, self.canvas.height)
        dye = getResolution(self.config["DYE_RESOLUTION"], self.canvas.width, self.canvas.height)
        sched = {"passes": _abi.SCHED_PASSES, "fused": _abi.SCHED_FUSED}[schedule]
        self.engines = [HipStripeEngine((sim["width"], sim["height"]), (dye["width"], dye["height"]), r // tiles_x, self.tiles_y,
                                        halo if world > 1 else 0, sched, device, part_x=r % tiles_x, parts_x=tiles_x, storage=storage)
                        for r in range(world)]
        for e in self.engines:
            e.use_own_stream()
            if reach is not None:
                e.set_reach(reach)
            if overlap is not None:
                e.set_overlap(overlap)
            if link_model is not None:
                e.set_link_model(*link_model)
        self.lib = _abi.lib()

    def close(self):
        for e in self.engines:
            e.close()

    def splat(self, x, y, dx, dy, color):
        r, g, b = (color["r"], color["g"], color["b"]) if isinstance(color, dict) else color
        aspect = self.canvas.width / self.canvas.height
        radius = self.config["SPLAT_RADIUS"] / 100.0
        if aspect > 1:
            radius *= aspect
        for e in self.engines:
            e.splat(x, y, dx, dy, r, g, b, aspect, radius)

    def multipleSplats(self, amount: int):
        for _ in range(int(amount)):
            c = HSVtoRGB(self.random(), 1.0, 1.0)
            color = {k: v * 0.15 * 10.0 for k, v in c.items()}
            x = self.random()
            y = self.random()
            dx = 1000 * (self.random() - 0.5)
            dy = 1000 * (self.random() - 0.5)
            self.splat(x, y, dx, dy, color)

    def step(self, dt: float, n: int = 1):
        c = self.config
        p = _abi.Params(float(c["CURL"]), float(c["PRESSURE"]), int(c["PRESSURE_ITERATIONS"]),
                        float(c["VELOCITY_DISSIPATION"]), float(c["DENSITY_DISSIPATION"]))
        arr = (C.c_void_p * self.world)(*[e.ctx.value for e in self.engines])
        rc = self.lib.fluid_group_step_n(arr, self.world, int(n), float(dt), C.byref(p))
        if rc != _abi.FLUID_OK:
            for e in self.engines:   # the failing context carries the message
                msg = self.lib.fluid_last_error(e.ctx)
                if msg:
                    raise _abi.FluidError(rc, msg.decode())
            _abi.check(None, rc)

    def sync(self):
        for e in self.engines:
            e.sync()

    def check_halo(self):
        for e in self.engines:
            e.check_halo()

    def read(self, name: str) -> np.ndarray:
        rows = [np.concatenate([self.engines[y * self.tiles_x + x].read(name) for x in range(self.tiles_x)], axis=1)
                for y in range(self.tiles_y)]
        return np.concatenate(rows, axis=0)

    @property
    def exchanges(self):
        return self.engines[0].exchange_count()


def run_local_stripes(world: int, body: Callable[["StripeSim"], object], **kw) -> List[object]:
    """Run `body(stripe_sim)` for every stripe in one process, one thread per stripe (LocalComm)."""
    import threading
    if kw.get("engine_factory") is None:
        import torch
        torch.cuda.init()  # initialise torch's HIP context once, on the main thread, before the stripe threads race for it
    hub = LocalComm.Hub(world)
    results: List[object] = [None] * world
    errors: List[BaseException] = []

    def worker(r):
        sim = None
        try:
            sim = StripeSim(comm=LocalComm(hub, r), **kw)
            results[r] = body(sim)
        except BaseException as ex:  # noqa: BLE001
            errors.append(ex)
        finally:
            if sim is not None:
                sim.close()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
