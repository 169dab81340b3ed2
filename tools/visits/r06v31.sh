#!/bin/bash
# round 6, visit 31: where a 16384 x 2048 rank at 200 iterations (configs[4]'s per-rank shape) spends its step: kernel timeline of one middle rank alone (loopback, instantaneous link)
OUT=$PWD/gpurun_out/r06v31; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( cd /tmp
FLUID_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so FAKE_RCCL_LOOPBACK=1 _OVL_CHILD='{"config": "deep16", "overlap": 1}' \
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o tr -- python $GRAFT_REPO_ROOT/tools/overlap_vs_link.py > $OUT/child.txt 2>$OUT/rocprof.err )
tail -1 $OUT/child.txt
F=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python - "$F" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("fluid::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "advect_both_fast_rects" in n]
lo = idx[-3] + 1 if len(idx) >= 3 else 0
hi = idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%6.1f gap  %7.1f us  %s  queue %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    prev_end = max(prev_end, e)
PY
head -150 $OUT/timeline.txt
rm -rf $OUT/prof
echo "== the single domain 16384^2 / 200 on this box =="
timeout 600 python bench.py --size 16384 --iters 200 --steps 6 --warmup 2 --cpu-budget 0 --no-traffic --no-steady --no-parity --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single 16384^2/200:', d['ms_per_step'], 'ms per step ->', d['ms_per_step']/8, 'per eighth')"
