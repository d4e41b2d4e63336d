#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/ppo_scan.sh <mult>... -- bench.py --cloud tum per TC_SURFACE_PPO_MULT (points per occupied cell of the
# adapted surface grid, relative to the default aim), no extras
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  TC_DEBUG=256 TC_SURFACE_PPO_MULT=$m timeout 300 python3 bench.py --cloud tum --steps 6 --warmup 2 --no-cpu-baseline --no-copy-probe --no-extras 2>gpurun_out/ppo.err | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mult $m', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'main us %.1f' % d['roofline']['avg_launch_us'], 'iteration us %.1f' % d['roofline']['iteration']['us'], 'normals Mpts/s %.0f' % d['normals_mpts_per_s'])"
  grep "index: n" gpurun_out/ppo.err | tail -1 | cut -c1-160
done
