#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/trace_tum.sh <tag> -- per-launch durations of the ICP kernels of the LAST 50-iteration call of
# `bench.py --cloud tum` (rocprofv3 --kernel-trace); TC_DEBUG / TC_HIP_LIB pass through
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_tum_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --cloud tum --steps 2 --warmup 1 --no-cpu-baseline --no-copy-probe --no-extras > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for pat in ("icp_correspond", "icp_refine"):
    d = [(e - s) / 1e3 for s, e, k in rows if pat in k][-50:]
    print(pat, "last call, us per launch:"); print("  " + " ".join(f"{x:.0f}" for x in d)); print(f"  mean {sum(d)/len(d):.1f}")
PY
