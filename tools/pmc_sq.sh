#!/bin/bash
# On the GPU box: SQ counters of the step's kernels (two passes of 8 counters), summarised per kernel.
TAG=${1:-sq}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 0 --cpu-budget 0 --no-profile-pass --no-traffic --no-steady --no-parity --settle-ms 0 >/dev/null 2>>$OUT/err.txt )
  F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && cp $F $OUT/pmc$i.csv
  rm -rf $OUT/p$i
done
python - $OUT/pmc1.csv $OUT/pmc2.csv <<'PY' | tee $OUT/sq_summary.txt
import csv, re, sys
from collections import defaultdict
def short(n):
    n = n.replace("fluid::(anonymous namespace)::", ""); n = re.sub(r"^void\s+", "", n); return re.sub(r"\(.*$", "", n)
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for path in sys.argv[1:]:
    try:
        rows = list(csv.DictReader(open(path)))
    except Exception as e:
        print("missing", path, e); continue
    for r in rows:
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, cs in agg.items():
    if not k.startswith("k_"): continue
    print(k)
    for c, v in sorted(cs.items()):
        n = max(len(cnt[(k, c)]), 1)
        print("   %-24s %16.0f per dispatch" % (c, v / n))
PY
