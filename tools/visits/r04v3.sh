#!/bin/bash
# round 4, visit 3: the whole GPU suite on the final build (compressed code objects, run-ahead below 1536^2), the overlap probe with enough
# hardware queues for the rank threads, then the round's canonical profile set (tools/gpu_round.sh quick).
OUT=gpurun_out/r04v3; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -x -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"
timeout 600 python tools/bench_single_step.py 1024 1536 2048 > $OUT/single_step.txt 2>&1; cat $OUT/single_step.txt | cut -c1-600
( timeout 700 python tools/overlap_vs_link.py --rounds 1 --config stripes2 --hw-queues 0
  timeout 700 python tools/overlap_vs_link.py --rounds 1 --config stripes2 --hw-queues 16
  timeout 700 python tools/overlap_vs_link.py --rounds 1 --config tiles2x2 --hw-queues 16
  timeout 700 python tools/overlap_vs_link.py --rounds 1 --config deep --hw-queues 16 ) > $OUT/overlap_vs_link_latency.txt 2>&1; cat $OUT/overlap_vs_link_latency.txt
bash tools/gpu_round.sh r04final quick > $OUT/gpu_round.log 2>&1; tail -40 $OUT/gpu_round.log | cut -c1-400
