#!/bin/bash
# Run ON THE GPU BOX: bash tools/pmc_passes.sh <tag> <script.py> -- per-unit PMC passes (SQ / TA / TCP / TCC)
# for one dev workload; each pass is its own rocprofv3 run (counters only, no trace domains).
set -u
TAG=$1; SCRIPT=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $SCRIPT > $OUT/log$i.txt 2>&1
done <<'PASSES'
SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TD_TD_BUSY_sum TD_TC_STALL_sum
PASSES
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        k = k.replace("void tc::", "").replace("tc::", "")
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
    if not any(s in k for s in ("icp_correspond", "icp_refine", "normals_knn", "normals_tagged")): continue
    print(k)
    for c, (v, n) in sorted(agg[k].items()):
        print(f"    {c:42s} {v/n/1e6:12.3f} M/launch  ({n} launches)")
PY
