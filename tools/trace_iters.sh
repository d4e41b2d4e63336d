#!/bin/bash
# Run ON THE GPU BOX: bash tools/trace_iters.sh <tag> -- per-launch durations of the ICP kernels of one
# 50-iteration call (rocprofv3 --kernel-trace), to see the moving vs the converged phase.
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 tools/dev/trace.py > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for pat in ("icp_correspond", "icp_refine", "icp_finalize"):
    d = [(e - s) / 1e3 for s, e, k in rows if pat in k]
    d = d[-50:]
    if not d: continue
    print(pat, "last call, us per launch:")
    print("  " + " ".join(f"{x:.1f}" if pat == "icp_finalize" else f"{x:.0f}" for x in d))
    print(f"  mean {sum(d)/len(d):.1f}   launches 2-37 {sum(d[1:37])/36:.1f}   launches 12-37 {sum(d[11:37])/26:.1f}   last 8 {sum(d[-8:])/8:.1f}")
starts = [s for s, e, k in rows if "icp_correspond" in k][-50:]
if len(starts) == 50:
    it = [(starts[i + 1] - starts[i]) / 1e3 for i in range(49)]
    print(f"iteration (start of a main pass to the start of the next), us: mean {sum(it)/len(it):.1f}   2-37 {sum(it[1:37])/36:.1f}   last 8 {sum(it[-9:-1])/8:.1f}")
PY
