#!/bin/bash
# TEST INFRASTRUCTURE: builds the in-process RCCL stand-in (see fake_rccl.cpp)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -o libfake_rccl.so fake_rccl.cpp -lpthread
echo "built $(pwd)/libfake_rccl.so"
