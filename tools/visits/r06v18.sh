#!/bin/bash
# round 6, visit 18: what one ordering primitive between two kernels costs the stream it sits in (tools/micro/sync_gap_probe.hip)
OUT=$PWD/gpurun_out/r06v18; mkdir -p $OUT
timeout 300 ./tools/micro/sync_gap_probe 2>&1 | tee $OUT/sync_gap_probe.txt
