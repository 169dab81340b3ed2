#!/bin/bash
# round 6, visit 35: the soak run at 4096^2 (the chained pressure launch, the packed dye) and at 3072^2; then the GPU suite once more in another order (-p no:randomly is not installed: reversed file order)
OUT=$PWD/gpurun_out/r06v35; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python tools/soak.py 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_4096.txt
timeout 900 python tools/soak.py 3072 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_3072.txt
timeout 1500 python -m pytest $(ls tests/test_*.py | sort -r) -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_gpu_reversed.txt
