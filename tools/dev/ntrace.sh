#!/bin/bash
# Run ON THE GPU BOX: kernel durations (rocprofv3 kernel trace, no events) of three estimate_normals(k = 16) calls on 1 M uniform points
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ntrace; mkdir -p gpurun_out/ntrace
cat > /tmp/ntrace.py <<'PY'
import sys; sys.path.insert(0, ".")
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1000000, 2)).cuda()
for rep in range(4): ctx.estimate_normals(d, 16)
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ntrace/t -- python3 /tmp/ntrace.py > gpurun_out/ntrace/log.txt 2>&1
python3 tools/dev/timeline.py gpurun_out/ntrace 14
