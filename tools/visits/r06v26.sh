#!/bin/bash
# round 6, visit 26: ranks on the general chained launch (panels + row / column ranges + rotation) against the single domain
OUT=$PWD/gpurun_out/r06v26; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_jacobi_chain.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
