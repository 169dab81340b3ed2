#!/bin/bash
# round 4, visit 17: the tile exchange in ONE round (diagonal neighbours), wide copy kernel, the first Jacobi launch as cover of the step's
# first exchange — parity (stripes / tiles / baseline sizes / node), then the link probe with and without the new cover
OUT=$PWD/gpurun_out/r04v17; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_node_shim.py tests/test_baseline_sizes.py tests/test_long_horizon.py -m gpu -q -x -rsx > $OUT/pytest_stripes.txt 2>&1; tail -5 $OUT/pytest_stripes.txt
timeout 700 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_new.txt 2>&1; cat $OUT/overlap_new.txt | cut -c1-160
FLUID_HIP_LIB=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so FLUID_COVER_JACOBI=0 timeout 700 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_cover_off.txt 2>&1; cat $OUT/overlap_cover_off.txt | cut -c1-160
