#!/bin/bash
# round 6, visit 16: the dye wire format agreed per call (all-reduce of four floats over the communicator), the chain rule (sets that fit the
# Infinity Cache, up to 24 blocks), schedule_info on ranks: the decomposition, chain and chain-safety tests
OUT=$PWD/gpurun_out/r06v16; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)  probes: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip_probes.so | cut -c1-16)"
timeout 1700 python -m pytest tests/test_stripes_gpu.py tests/test_jacobi_chain.py tests/test_chain_safety.py tests/test_baseline_sizes.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest.txt
