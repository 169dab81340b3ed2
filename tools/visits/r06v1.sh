#!/bin/bash
# round 6, visit 1: this round's box before anything changes — (a) the driver's command twice and the default bench once (the A/B reference),
# (b) where the dispatcher puts workgroups and waves (tools/micro/placement_probe.hip), (c) does FETCH_SIZE count Infinity-Cache hits
# (tools/micro/mall_probe.hip), (d) what bounds the fused advection: TA / TCP / TCC / SQ counters of k_advect_both_fast_rgb (item 4).
OUT=$PWD/gpurun_out/r06v1; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "lib: $(sha256sum webgl-fluid-simulation_amd/libfluid_hip.so | cut -c1-16)"
rocprofv3 -L > $OUT/rocprofv3_counters.txt 2>&1
grep -c . $OUT/rocprofv3_counters.txt
echo "== (a) driver command x2, default bench =="
for k in 1 2; do timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/driver_cmd_run$k.json 2>>$OUT/bench.err; python - $OUT/driver_cmd_run$k.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver cmd: ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "roofline.frac", (d.get("roofline") or {}).get("frac"), "avg_launch_ms", (d.get("roofline") or {}).get("avg_launch_ms"), "err", d.get("error"))
PY
done
FLUID_BENCH_KEEP_PMC="$OUT" timeout 1200 python bench.py > $OUT/bench.json 2>>$OUT/bench.err; tail -c 3000 $OUT/bench.json
echo "== (b) placement probe =="
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/placement_probe tools/micro/placement_probe.hip && { /tmp/placement_probe 3000 15; /tmp/placement_probe 6210 30; } | tee $OUT/placement_probe.txt
echo "== (c) MALL probe =="
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/micro/mall_probe.hip && /tmp/mall_probe 20 | tee $OUT/mall_probe.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/mp_$C -o pmc -- /tmp/mall_probe 5 > /dev/null 2>>$OUT/err.txt )
  F=$(find $OUT/mp_$C -name '*counter_collection.csv' | head -1); [ -n "$F" ] && cp $F $OUT/mall_probe_$C.csv; rm -rf $OUT/mp_$C
done
python - $OUT/mall_probe_FETCH_SIZE.csv $OUT/mall_probe_WRITE_SIZE.csv <<'PY' | tee -a $OUT/mall_probe.txt
import csv, sys, re
from collections import defaultdict
for path in sys.argv[1:]:
    try: rows = list(csv.DictReader(open(path)))
    except Exception as e: print("missing", path, e); continue
    per = defaultdict(float); name = {}
    for r in rows:
        per[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "")); c = r["Counter_Name"]
    agg = defaultdict(list)
    for d, v in per.items(): agg[name[d]].append(v)
    for k in sorted(agg): print("%-18s %-11s per dispatch: last %10.1f KiB (raw), all: %s" % (k, c, agg[k][-1], " ".join("%.0f" % x for x in agg[k])))
PY
echo "== (d) advection counters =="
bash tools/pmc_kernel.sh r06v1/adv k_advect_both_fast \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" 2>&1 | tail -80
echo "== done =="
