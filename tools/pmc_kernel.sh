#!/bin/bash
# On the GPU box: hardware counters of ONE kernel of the headline step, as many passes as counter sets given (each set must fit the
# per-block slot limits: SQ 8, TCC 4, TA/TCP a few each — MI355X_MICROARCH.md "rocprofv3 PMC slots"); per-dispatch averages, summed over
# the XCD / SE instances.  Usage: tools/pmc_kernel.sh <tag> <kernel name substring> "<set 1>" ["<set 2>" ...]   [BENCH_ARGS=... in the environment]
TAG=$1; KERN=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for P in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --cpu-budget 0 --no-profile-pass --no-traffic --no-steady --no-parity --settle-ms 0 ${BENCH_ARGS:-} >/dev/null 2>>$OUT/err.txt )
  F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then cp $F $OUT/pmc$i.csv; else echo "set $i produced no counter file: $P" | tee -a $OUT/err.txt; fi
  rm -rf $OUT/p$i
done
python - "$KERN" $OUT/pmc*.csv <<'PY' | tee $OUT/summary.txt
import csv, re, sys
from collections import defaultdict
kern = sys.argv[1]
def short(n):
    n = n.replace("fluid::(anonymous namespace)::", ""); n = re.sub(r"^void\s+", "", n); return re.sub(r"\(.*$", "", n)
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set); grid = {}
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if kern not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
        grid[k] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"))
for k, cs in agg.items():
    print(k, "grid/wg/vgpr/agpr/sgpr/lds:", grid[k])
    for c, v in sorted(cs.items()):
        n = max(len(cnt[(k, c)]), 1)
        print("   %-40s %18.0f per dispatch  (%d dispatches)" % (c, v / n, n))
PY
