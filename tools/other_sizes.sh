#!/bin/bash
# GPU box: bench.py at the other BASELINE.json grid sizes (configs[1] 1024^2, configs[3]'s 8192^2, configs[4]'s 16384^2 / 200 iterations)
# with the HBM traffic measured in the run, then the soak run.  Usage: bash tools/other_sizes.sh <tag>
OUT=gpurun_out/${1:-other_sizes}; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; s = d.get('step_hbm') or {}
print('size %5d iters %3d: %9.2f steps/s  %6.2f GLUPS  %8.4f ms/step | %s %.1f us/launch, %.1f MB/launch (%s) -> roofline.frac %.3f | step %.2f GB -> %.3f of 8 TB/s' % (
    $1, $2, d['steps_per_sec'], d['value'], d['ms_per_step'], r['kernel'].split('<')[0], r['avg_launch_ms'] * 1e3, r['traffic'] / 1e6,
    'PMC' if 'PMC' in r['traffic_source'] else 'model', r['frac'], s.get('bytes_per_step', 0) / 1e9, s.get('frac', 0)))
"; }
echo "# bench.py --size N --iters I on one MI355X (fused schedule, dye grid = sim grid), HBM bytes from PMC passes inside each run" | tee $OUT/bench_other_sizes.txt
timeout 300 python bench.py --size 1024 --iters 50 --steps 2000 --warmup 200 --cpu-budget 0 --no-steady --no-parity 2>>$OUT/err.txt | line 1024 50 | tee -a $OUT/bench_other_sizes.txt
timeout 300 python bench.py --size 2048 --iters 50 --steps 800 --warmup 100 --cpu-budget 0 --no-steady --no-parity 2>>$OUT/err.txt | line 2048 50 | tee -a $OUT/bench_other_sizes.txt
timeout 400 python bench.py --size 8192 --iters 50 --steps 100 --warmup 20 --cpu-budget 0 --no-steady --no-parity 2>>$OUT/err.txt | line 8192 50 | tee -a $OUT/bench_other_sizes.txt
timeout 600 python bench.py --size 16384 --iters 200 --steps 20 --warmup 4 --cpu-budget 0 --no-steady --no-parity 2>>$OUT/err.txt | line 16384 200 | tee -a $OUT/bench_other_sizes.txt
echo "== soak ==" | tee $OUT/soak.txt
timeout 400 python tools/soak.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a $OUT/soak.txt
