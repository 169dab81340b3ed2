"""The pressure loop as ONE launch of chained blocks of iterations (k_jacobi_tb_chain, round 5; any width and up to 24 blocks since round 6):
fp32 grids of 3072^2 ... 20 M texels — those whose pressure set fits the Infinity Cache — run the step's 50 Jacobi iterations (pressureShader script.js:868-890, loop 1259-1266) as one grid of 5 x T workgroups in which a tile of block l
waits for the three tile rows of block l - 1 around it — no fill / drain between the blocks.  Same iterations over the same texels, hence the
same bits: held here to the one-kernel-per-pass schedule (which the goldens pin to the live reference) on the shapes the rule selects, the
iteration counts that cut unevenly, and next to the shapes it must leave alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 0.016666


@pytest.mark.parametrize("w,h,iters,chained", [
    (4096, 4096, 50, True),      # the headline
    (4096, 3072, 47, True),      # non-square, blocks of 10 / 10 / 9 / 9 / 9
    (4200, 3000, 11, True),      # W % 4 == 0 but no power of two; two blocks (6 + 5)
    (3800, 2600, 80, True),      # the narrow end of the rule (17 tiles per row), eight blocks
    (4096, 3072, 200, True),     # configs[4]'s iteration count: twenty blocks in one launch (round 6: the limit was eight; -8.5 % of the 4096^2 step)
    (4096, 2560, 137, True),     # fourteen blocks of 10 / 9
    (4096, 2560, 250, False),    # beyond twenty-four blocks: plain launches
    (4096, 4096, 10, False),     # one block: a plain launch
    (4096, 2048, 50, False),     # below 3072^2 texels: the small-grid tile with the gradient subtract folded in, five launches
    (3072, 3072, 50, True),      # four rows of tiles per XCD band (round 6, with the bands rotating over the XCDs: -12 % of the loop; it was level)
    (6144, 2048, 50, True),      # a tile row longer than an XCD holds: two panels of 14 tiles (round 6)
    (8192, 2048, 50, True),      # 35 tile rows = 12 bands of 3: the shape whose fixed XCD assignment measured +10 %; rotated -4 %
    (16384, 1024, 33, True),     # four panels, 18 tile rows
    (2048, 8192, 50, True),      # nine tiles per row, seven rows per band
    (4096, 5120, 50, False),     # 21 M texels: the loop's set (252 MB) no longer fits the Infinity Cache — plain launches
    (4096, 8192, 50, False),     # (round 5's rule chained this one: +4.5 %)
])
def test_chained_pressure_loop_leaves_the_same_bits(w, h, iters, chained):
    import fluid_hip
    cfg = {"SIM_RESOLUTION": min(w, h), "DYE_RESOLUTION": min(w, h), "PRESSURE_ITERATIONS": iters}
    sims = [fluid_hip.FluidSim(canvas=(w, h), config=cfg, schedule=s, random=fluid_hip.mulberry32(21)) for s in ("passes", "fused")]
    try:
        info = sims[1].schedule_info(3, DT)
        assert bool(info["jacobi_chained"]) is chained and info["jacobi_launches"] == -(-iters // 10), info
        if chained:   # curl / vorticity / divergence, the ONE launch of the loop, gradient subtract, advection
            assert info["launches"] == 3 * 4, info
        for s in sims:
            s.multipleSplats(5)
            s.step(DT, 2)
            s.multipleSplats(1)
            s.step(DT, 1)            # a call of one step: the same loop again
        for k in ("velocity", "pressure", "divergence", "curl", "dye"):
            assert np.array_equal(sims[0].read(k), sims[1].read(k)), k
        # the loop on its own, through the per-pass entry point (what the stripe driver's hosted form and the tests call)
        for s in sims:
            s.run_pass("jacobi", iters=iters)
        assert np.array_equal(sims[0].read("pressure"), sims[1].read("pressure"))
    finally:
        for s in sims:
            s.close()


def test_a_long_run_through_the_chained_loop_stays_bitwise():
    """300 steps at the headline size in calls of 1, 7 and 50 steps: fused (chained loop, packed dye) == per-pass after every burst"""
    import fluid_hip
    cfg = {"SIM_RESOLUTION": 4096, "DYE_RESOLUTION": 4096, "PRESSURE_ITERATIONS": 50}
    sims = [fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, schedule=s, random=fluid_hip.mulberry32(3)) for s in ("passes", "fused")]
    try:
        for s in sims:
            s.multipleSplats(12)
        for n in (1, 7, 50, 1, 50, 50, 50, 41, 50):
            for s in sims:
                s.step(DT, n)
            for k in ("velocity", "pressure", "dye"):
                assert np.array_equal(sims[0].read(k), sims[1].read(k)), (n, k)
    finally:
        for s in sims:
            s.close()


@pytest.mark.parametrize("canvas,world,tiles_x,iters,overlap", [
    ((8192, 4096), 2, 1, 50, True),      # two 8192 x 2048 stripe ranks: 36 tiles per row = two panels, a row range per block; the single domain (33 M texels) runs plain launches
    ((8192, 4096), 2, 1, 130, False),    # thirteen blocks, no cut launches in front of them
    ((12288, 2048), 2, 2, 50, True),     # two 6144 x 2048 column tiles: panels of 14 tiles AND a column range per block
    ((6144, 6144), 4, 2, 33, True),      # 2 x 2 tiles of 3072^2: the smallest grid the rule takes, eight neighbours' ghost texels around it
])
def test_ranks_run_the_general_chained_launch_and_leave_the_single_domains_bits(canvas, world, tiles_x, iters, overlap):
    """Round 6: the chained launch at any width (panels), with the bands rotating over the XCDs, on stripe / tile RANKS — every block of a rank
    has its own row / column range (a rank recomputes fewer ghost texels from launch to launch), tiles with nothing to store only count
    themselves.  An in-process set against the single domain of the whole grid, bit for bit; at these sizes the single domain does not chain
    (its set does not fit the Infinity Cache) or chains with other panels — either way a different launch structure leaves the same bits."""
    import fluid_hip
    from fluid_hip.stripes import StripeGroup
    res = min(canvas)
    cfg = {"SIM_RESOLUTION": res, "DYE_RESOLUTION": res, "PRESSURE_ITERATIONS": iters}
    g = StripeGroup(world, canvas=canvas, config=cfg, halo=56, random=fluid_hip.mulberry32(31), tiles_x=tiles_x, overlap=overlap)
    try:
        with fluid_hip.FluidSim(canvas=canvas, config=cfg, random=fluid_hip.mulberry32(31)) as one:
            for sim in (g, one):
                sim.multipleSplats(5)
                sim.step(DT, 2)
            info = [e.schedule_info(2, DT, g.config) for e in g.engines]
            assert all(i["jacobi_chained"] for i in info), info
            g.check_halo()
            for k in ("velocity", "pressure", "divergence", "curl", "dye"):
                assert np.array_equal(g.read(k), one.read(k)), k
    finally:
        g.close()
