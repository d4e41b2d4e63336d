#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/ab_quick.sh <variant>... -- bench (no extras) with the default library and each variant, two rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in default "$@"; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  TC_HIP_LIB=$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-copy-probe --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'main us %.1f' % d['roofline']['avg_launch_us'], 'normals Mpts/s %.0f' % d['normals_mpts_per_s'])"
done; done
