#!/bin/bash
# round 6, visit 34: did the lab form of visit 33 (two more kernel arguments, a larger ChainPlan) cost the PRODUCT anything?  The product library of commit a60674a
# (build_ab/old) against this tree's (build_ab/new), same flags, interleaved, default bench window
OUT=$PWD/gpurun_out/r06v34; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
sha256sum build_ab/old/libfluid_hip.so build_ab/new/libfluid_hip.so webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16,65- | tee $OUT/ab_old_new.txt
timeout 900 python tools/ab_env.py --rounds 5 --args "--steps 200 --warmup 50 --no-profile-pass --no-parity" "FLUID_HIP_LIB=/root/repo/build_ab/old/libfluid_hip.so" "FLUID_HIP_LIB=/root/repo/build_ab/new/libfluid_hip.so" 2>&1 | tee -a $OUT/ab_old_new.txt
