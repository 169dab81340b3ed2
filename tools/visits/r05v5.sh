#!/bin/bash
# round 5, visit 5: the dye != sim dye pass.  Round 4's velocity-tile kernels read their wave's LDS tile with FLAT loads (28 per thread at four rows:
# the compiler merged the "tap in the tile" / "tap gathered" branches into one load through a generic pointer) — texture-addresser instructions in the
# kernel the addresser bounds.  Now typed LDS pointers -> ds_read2_b64.  Plus the lab variant that stages the wave's DYE tap box through LDS (FLUID_DYE_BOX=1).
OUT=$PWD/gpurun_out/r05v5; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
echo "== parity of everything that advects on two grids =="
timeout 1200 python -m pytest tests/test_hip_vs_golden.py tests/test_hip_vs_oracle.py tests/test_hip_properties.py tests/test_long_horizon.py tests/test_input_replay.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.txt
echo "== sim 1024 / dye 4096 / 20: this tree (ds_read velocity tile) against the dye tap box (lab) and the RGBA kernel =="
timeout 600 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 3 "FLUID_SKIP_CURL=1" "FLUID_DYE_BOX=1" "FLUID_DYE_PACK=0" 2>&1 | tee $OUT/dye_ne_sim_1024_4096.txt
echo "== sim 2048 / dye 8192 / 20 =="
timeout 600 python tools/ab_passes.py --sim 2048 --dye 8192 --iters 20 --rounds 2 --steps 200 "FLUID_SKIP_CURL=1" "FLUID_DYE_BOX=1" 2>&1 | tee $OUT/dye_ne_sim_2048_8192.txt
echo "== sim 256 / dye 2048 / 20 (the reference's ratio 8) =="
timeout 600 python tools/ab_passes.py --sim 256 --dye 2048 --iters 20 --rounds 2 "FLUID_SKIP_CURL=1" "FLUID_DYE_BOX=1" 2>&1 | tee $OUT/dye_ne_sim_256_2048.txt
echo "== the reference's shipping configuration (sim 128 / dye 1024 / 20) =="
timeout 300 python tools/bench_shipping.py 2>&1 | tee $OUT/bench_shipping_defaults.json
echo "== one stripe rank, 200-iteration regime: frames on the comm stream only with one cut launch =="
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
FLUID_HIP_LIB=$PROBES FLUID_SKIP_CURL=1 timeout 600 python tools/overlap_vs_link.py --config deep --quick --rounds 2 2>&1 | tee $OUT/rank_deep_r05.txt
FLUID_HIP_LIB=$PROBES FLUID_STRIPS_ON_COMM=0 FLUID_DYE_PACK=0 timeout 600 python tools/overlap_vs_link.py --config deep --quick --rounds 1 2>&1 | tee $OUT/rank_deep_r04.txt
echo "== 2 x 2 tiles as an in-process group, overlap off (the unpack per step of visit 4 is gone) =="
timeout 600 python tools/bench_group.py 4096 50 56 4 2 2>&1 | tail -2 | tee $OUT/group_tiles.txt
