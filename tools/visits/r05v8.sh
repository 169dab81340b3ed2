#!/bin/bash
# round 5, visit 8: (a) the kernel timeline of ONE stripe rank under this round's schedule (loopback, instantaneous link) — round 4's is
# profiles/r04/stripe_rank_timeline.txt; (b) how repeatable the link calibration is (three synthetic links, three probes each);
# (c) the driver's command seven more times (lease 4).
OUT=$PWD/gpurun_out/r05v8; mkdir -p $OUT
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
head -70 $OUT/timeline.txt
rm -rf $OUT/prof
echo "== link calibration, three probes per synthetic link =="
for link in "40 25" "80 50" "150 100" "10 200"; do set -- $link
  for k in 1 2 3; do
    FLUID_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so FAKE_RCCL_LOOPBACK=1 FAKE_RCCL_DELAY_US=$1 FAKE_RCCL_GBPS=$2 _OVL_CHILD='{"config": "stripe", "overlap": 1, "calibrate": true}' \
      timeout 120 python tools/overlap_vs_link.py 2>/dev/null | grep latency | sed "s/^/injected $1 us + bytes \/ $2 GB\/s -> /"
  done
done | tee $OUT/link_calibration_repeat.txt
bash tools/visits/r05_driver_lease.sh 4
