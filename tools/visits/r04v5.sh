#!/bin/bash
# round 4, visit 5: the tile exchange with its two rounds side by side + corner round, the Jacobi blocks behind pressure-only exchanges cut
# interior-first: parity (stripe / tile / baseline-size tests), then the one-rank loopback probe again; L1 / L2 request counters of the
# advection kernels.
OUT=$PWD/gpurun_out/r04v5; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_stripes_gpu.py tests/test_baseline_sizes.py tests/test_long_horizon.py tests/test_hip_f16.py tests/test_node_shim.py -m gpu -q -rsx > $OUT/pytest_stripes.txt 2>&1; tail -6 $OUT/pytest_stripes.txt
timeout 1500 python tools/overlap_vs_link.py --rounds 1 > $OUT/overlap_vs_link_latency.txt 2>&1; cat $OUT/overlap_vs_link_latency.txt
for P in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  T=$(echo $P | cut -c1-14)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/c_$T -o pmc -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT/webgl-fluid-simulation_amd')
import fluid_hip
for sim_res, dye_res in ((1024, 4096), (4096, 4096)):
    cfg = {'SIM_RESOLUTION': sim_res, 'DYE_RESOLUTION': dye_res, 'PRESSURE_ITERATIONS': 20}
    with fluid_hip.FluidSim(canvas=(4096, 4096), config=cfg, random=fluid_hip.mulberry32(1234)) as sim:
        sim.multipleSplats(10); sim.step(0.016666, 60); sim.sync()
" > /dev/null 2>> $OUT/counters.err )
  F=$(find $OUT/c_$T -name '*counter_collection.csv' | head -1); [ -n "$F" ] && cp $F $OUT/counters_$T.csv; rm -rf $OUT/c_$T
done
tail -3 $OUT/counters.err
python - <<'PY'
import csv, glob, re, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04v5/counters_*.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("fluid::(anonymous namespace)::", "").replace("void ", ""))
        if k.startswith("k_advect") or k.startswith("k_gradsub4") or k.startswith("k_curl") or k.startswith("k_jacobi_tb_mix"):
            agg[k + " grid=" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-38s %16.0f per dispatch (%d)" % (c, sum(v) / len(v), len(v)))
PY
