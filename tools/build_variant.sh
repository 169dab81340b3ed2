#!/bin/bash
# Build another copy of libfluid_hip.so with extra compiler flags, in-tree (so that it travels with the gpurun snapshot):
#   bash tools/build_variant.sh <name> "<extra flags>"   ->  build_ab/<name>/libfluid_hip.so   (load with FLUID_HIP_LIB=...)
set -e
NAME=$1; FL=$2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D=$ROOT/build_ab/$NAME
rm -rf "$D"; mkdir -p "$D"
cp -r "$ROOT/webgl-fluid-simulation_amd/csrc" "$D/csrc"; cp "$ROOT/webgl-fluid-simulation_amd/Makefile" "$D/Makefile"
mkdir -p "$ROOT/build_ab/include"; cp "$ROOT/include/"*.h "$ROOT/build_ab/include/"
make -C "$D" -j4 EXTRA="$FL" 2>&1 | grep -E "error|warning: unused" | head -5 || true
ls -la "$D/libfluid_hip.so"
