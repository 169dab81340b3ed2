#!/bin/bash
# round 4, visit 20: up to two cut Jacobi launches as cover (link model picks how many) — parity, then the probe: product default /
# minimal cover (model says the link is instantaneous) / maximal cover (model says it is slow), three rounds each, alternating
OUT=$PWD/gpurun_out/r04v20; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_long_horizon.py -m gpu -q -x -rsx > $OUT/pytest_stripes.txt 2>&1; tail -5 $OUT/pytest_stripes.txt
P=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for r in 1 2 3; do
  timeout 400 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_default_$r.txt 2>&1
  FLUID_HIP_LIB=$P FLUID_LINK_MODEL=0,1000 timeout 400 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_min_$r.txt 2>&1
  FLUID_HIP_LIB=$P FLUID_LINK_MODEL=200,10 timeout 400 python tools/overlap_vs_link.py --quick --rounds 1 > $OUT/overlap_max_$r.txt 2>&1
done
for v in default min max; do for r in 1 2 3; do echo "== $v $r"; grep "ms/step" $OUT/overlap_${v}_$r.txt | cut -c1-62; done; done
