#!/bin/bash
# Builds the CPU oracle (test infrastructure) -> oracle/libfluid_oracle.so
# -ffp-contract=off: the reference's shader arithmetic is unfused fp32.
set -e
cd "$(dirname "$0")"
gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC -o libfluid_oracle.so fluid_oracle.c -lm
echo "built $(pwd)/libfluid_oracle.so"
