#!/bin/bash
# round 4, visit 29: kernel timeline of ONE stripe rank (loopback, instantaneous link): where do the +12 % over the single domain go?
OUT=$PWD/gpurun_out/r04v29; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
FLUID_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so FAKE_RCCL_LOOPBACK=1 _OVL_CHILD='{"config": "stripe", "overlap": 1}' \
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o tr -- python $GRAFT_REPO_ROOT/tools/overlap_vs_link.py > $OUT/child.txt 2>$OUT/rocprof.err
cat $OUT/child.txt | tail -2
F=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python - "$F" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 3 steps: find the last advect strips launches
names = [r["Kernel_Name"] for r in rows]
def short(n):
    n = n.replace("fluid::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]
idx = [i for i, n in enumerate(names) if "advect_both_fast_rects" in n]
lo = idx[-4] + 1 if len(idx) >= 4 else 0
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%6.1f gap  %7.1f us  %s  grid %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size", "?")))
    prev_end = max(prev_end, e)
PY
cat $OUT/timeline.txt | head -80
rm -rf $OUT/prof
