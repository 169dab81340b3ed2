#!/bin/bash
# round 2, visit 11: the full suite with RCCL logging, to see where the node tile-rank process sits when it hangs
set -u
OUT=$PWD/gpurun_out/r02_v11; mkdir -p $OUT /tmp/nccl_logs
export TMPDIR=/tmp PYTHONUNBUFFERED=1
export NCCL_DEBUG=INFO NCCL_DEBUG_FILE=/tmp/nccl_logs/nccl_%p.log
( while true; do sleep 20; date +%T; rocm-smi --showmemuse --showuse 2>/dev/null | grep -E "GPU\[0\]" | head -3; ps -eo pid,etime,cmd | grep -E "node |run_ranks" | grep -v grep | cut -c1-120; done ) > $OUT/monitor.txt 2>&1 &
MON=$!
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/log.txt
kill $MON
echo "== nccl logs ==" | tee -a $OUT/log.txt
ls -la /tmp/nccl_logs | tail -8 | tee -a $OUT/log.txt
for f in $(ls -t /tmp/nccl_logs | head -2); do echo "--- $f"; tail -25 /tmp/nccl_logs/$f; done | cut -c1-220 | tee -a $OUT/log.txt
tail -30 $OUT/monitor.txt | tee -a $OUT/log.txt
echo "== done ==" | tee -a $OUT/log.txt
