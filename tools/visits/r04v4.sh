#!/bin/bash
# round 4, visit 4: the whole GPU suite (visit 3's stopped at a wrong assertion of the live bench test), the overlap probe as ONE rank in
# loopback, the advection blocks with their waves stacked in y (lab knob FLUID_ADVECT_WY) on both the 4096^2 headline and the dye != sim case,
# and the L1 / L2 request counters of the dye != sim advection.
OUT=gpurun_out/r04v4; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -rsx > $OUT/pytest_gpu.txt 2>&1; tail -6 $OUT/pytest_gpu.txt
timeout 1500 python tools/overlap_vs_link.py --rounds 1 > $OUT/overlap_vs_link_latency.txt 2>&1; cat $OUT/overlap_vs_link_latency.txt
timeout 900 python tools/ab_env.py --rounds 2 --args "--steps 200 --warmup 50 --no-parity" "FLUID_SKIP_CURL=1" "FLUID_ADVECT_WY=2" "FLUID_ADVECT_WY=4" "FLUID_ADVECT_WY=4 FLUID_ADVECT_ROWS=2" "FLUID_ADVECT_WY=2 FLUID_ADVECT_ROWS=2" > $OUT/ab_advect_wy_4096.txt 2>&1; cat $OUT/ab_advect_wy_4096.txt
timeout 900 python tools/ab_passes.py --sim 1024 --dye 4096 --iters 20 --rounds 2 "FLUID_SKIP_CURL=1" "FLUID_ADVECT_WY=2" "FLUID_ADVECT_WY=4" "FLUID_ADVECT_WY=4 FLUID_ADVECT_SPLIT_ROWS=4" "FLUID_ADVECT_WY=2 FLUID_ADVECT_SPLIT_ROWS=4" "FLUID_ADVECT_SPLIT_ROWS=1" > $OUT/ab_dye_ne_sim.txt 2>&1; cat $OUT/ab_dye_ne_sim.txt
rocprofv3 -L 2>/dev/null | grep -i "TCP_\|TA_\|TCC_REQ\|TCC_HIT\|TCC_MISS\|TCC_EA0_RD" | cut -c1-160 | head -80 > $OUT/counters_available.txt; wc -l $OUT/counters_available.txt
for P in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  T=$(echo $P | cut -c1-12)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/c_$T -o pmc -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT/webgl-fluid-simulation_amd')
import fluid_hip
for sim_res, dye_res in ((1024, 4096), (4096, 4096)):
    cfg = {'SIM_RESOLUTION': sim_res, 'DYE_RESOLUTION': dye_res, 'PRESSURE_ITERATIONS': 20}
    with fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
        sim.multipleSplats(10); sim.step(0.016666, 6); sim.sync()
" > /dev/null 2>> $OUT/counters.err )
  F=$(find $OUT/c_$T -name '*counter_collection.csv' | head -1); [ -n "$F" ] && cp $F $OUT/counters_$T.csv; rm -rf $OUT/c_$T
done
python - <<'PY'
import csv, glob, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04v4/counters_*.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("fluid::(anonymous namespace)::", "").replace("void ", ""))
        if k.startswith("k_advect") or k.startswith("k_gradsub4") or k.startswith("k_curl"):
            agg[k + " grid=" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-34s %16.0f per dispatch (%d)" % (c, sum(v) / len(v), len(v)))
PY
