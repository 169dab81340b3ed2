#!/bin/bash
# round 6, visit 17: where a stripe rank stands against the single domain on this round's tree (chained loop on both), its kernel timeline, and
# the lab's device-scope events
OUT=$PWD/gpurun_out/r06v17; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)"
( cd /tmp
FLUID_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so FAKE_RCCL_LOOPBACK=1 _OVL_CHILD='{"config": "stripe", "overlap": 1}' \
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o tr -- python $GRAFT_REPO_ROOT/tools/overlap_vs_link.py > $OUT/child.txt 2>$OUT/rocprof.err )
tail -2 $OUT/child.txt
F=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python - "$F" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
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
    print("%9.1f us  +%6.1f gap  %7.1f us  %s  queue %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    prev_end = max(prev_end, e)
PY
head -60 $OUT/timeline.txt
rm -rf $OUT/prof
PROBES=$PWD/webgl-fluid-simulation_amd/libfluid_hip_probes.so
for k in 1 2; do
echo "== one stripe rank alone, product library =="
timeout 600 python tools/overlap_vs_link.py --config stripe --quick --rounds 1 2>&1 | grep "link   0\|link  60" | tee -a $OUT/rank_stripe.txt
echo "== ... lab build, FLUID_EVENT_SCOPE=device =="
FLUID_HIP_LIB=$PROBES FLUID_EVENT_SCOPE=device timeout 600 python tools/overlap_vs_link.py --config stripe --quick --rounds 1 2>&1 | grep "link   0\|link  60" | tee -a $OUT/rank_stripe_devscope.txt
echo "== the single domain, same box =="
timeout 300 python tools/ab_env.py --rounds 1 --args "--steps 100 --warmup 40 --no-profile-pass --no-parity" "" 2>&1 | tee -a $OUT/single_4096.txt
done
