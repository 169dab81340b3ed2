"""The multi-GPU stripe driver (fluid_hip.stripes.StripeSim) on CPU: the decomposition, the ghost-row
validity bookkeeping and the exchange sequencing, with the CPU oracle injected as the compute engine.
Criterion: the decomposed result is BITWISE equal to the whole-domain oracle run.
  * LocalComm: all stripes in one process (threads) — worlds 2, 3, 4
  * TorchDistComm over gloo, world_size 2, one process per rank — the path bench.py --gpus N takes"""
import os
import socket

import numpy as np
import pytest

import scenario as S
from oracle_engine import OracleStripeEngine

CASES = [
    # canvas, config, halo, world, steps
    ((256, 256), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 20}, 8, 2, 3),
    ((256, 256), {"SIM_RESOLUTION": 96, "DYE_RESOLUTION": 96, "PRESSURE_ITERATIONS": 50}, 12, 3, 2),   # iters > halo: 6 pressure exchanges
    ((256, 256), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 128, "PRESSURE_ITERATIONS": 10}, 10, 4, 2),  # dye res != sim res
    ((200, 400), {"SIM_RESOLUTION": 40, "DYE_RESOLUTION": 40, "PRESSURE_ITERATIONS": 0, "CURL": 0}, 8, 2, 2),  # no Jacobi at all
    ((256, 256), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 7}, 16, 4, 2),    # halo == stripe height
]


def whole_domain(oracle, canvas, cfg, steps, seed=77):
    ref = oracle.RefSim(canvas=canvas, config=cfg, seed=seed)
    ref.multiple_splats(5)
    ref.step(0.016666, steps)
    return ref.fields()


@pytest.mark.parametrize("canvas,cfg,halo,world,steps", CASES)
def test_local_stripes_bitwise(oracle, canvas, cfg, halo, world, steps):
    import fluid_hip
    from fluid_hip.stripes import run_local_stripes
    want = whole_domain(oracle, canvas, cfg, steps)

    def body(sim):
        sim.multipleSplats(5)
        for _ in range(steps):
            sim.step(0.016666)
        sim.check_halo()
        return {k: sim.read_local(k) for k in S.FIELDS}, sim.exchanges

    # every rank must draw the same splat stream: each gets its own generator with the same seed
    def body_seeded(sim):
        sim.random = fluid_hip.mulberry32(77)
        return body(sim)

    res = run_local_stripes(world, body_seeded, canvas=canvas, config=cfg, halo=halo, engine_factory=OracleStripeEngine)
    for k in S.FIELDS:
        got = np.concatenate([r[0][k] for r in res], axis=0)
        assert got.shape == want[k].shape
        assert np.array_equal(got, want[k]), k
    iters = cfg["PRESSURE_ITERATIONS"]
    blocks = max(1, -(-iters // (halo - 3)))
    assert res[0][1] == steps * (2 + max(blocks - 1, 0))   # {velocity, pressure}, pressure per further Jacobi block, {velocity, dye}


def test_halo_overflow_is_detected(oracle):
    import fluid_hip
    from fluid_hip.stripes import run_local_stripes
    cfg = {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 64, "PRESSURE_ITERATIONS": 4}

    def body(sim):
        sim.random = fluid_hip.mulberry32(1)
        sim.splat(0.5, 0.5, 0.0, 90000.0, (1, 1, 1))   # dt*|v| >> 4 rows
        sim.step(0.016666)
        sim.check_halo()

    with pytest.raises(RuntimeError):
        run_local_stripes(2, body, canvas=(256, 256), config=cfg, halo=4, engine_factory=OracleStripeEngine)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, canvas, cfg, halo, steps, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (root, os.path.join(root, "webgl-fluid-simulation_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import fluid_hip
    from fluid_hip.stripes import StripeSim
    from oracle_engine import OracleStripeEngine as Eng
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sim = StripeSim(canvas=canvas, config=cfg, halo=halo, random=fluid_hip.mulberry32(77), engine_factory=Eng)
        sim.multipleSplats(5)
        for _ in range(steps):
            sim.step(0.016666)
        sim.check_halo()
        full = {k: sim.read(k) for k in ("velocity", "pressure", "divergence", "curl", "dye")}
        if rank == 0:
            np.savez(os.path.join(out_dir, "stripes.npz"), **full)
    finally:
        dist.destroy_process_group()


def test_gloo_world2_bitwise(oracle, tmp_path):
    import torch.multiprocessing as mp
    canvas, cfg, halo, steps = (256, 256), {"SIM_RESOLUTION": 64, "DYE_RESOLUTION": 128, "PRESSURE_ITERATIONS": 20}, 8, 2
    want = whole_domain(oracle, canvas, cfg, steps)
    mp.spawn(_gloo_worker, args=(2, _free_port(), canvas, cfg, halo, steps, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "stripes.npz"))
    for k in S.FIELDS:
        assert np.array_equal(got[k], want[k]), k
