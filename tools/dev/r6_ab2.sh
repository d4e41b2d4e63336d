#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_ab2.sh <variant>... -- bench A/B, 4 rounds x 20 steps, uniform and TUM-shaped, default library against variants
cd "$GRAFT_REPO_ROOT"
for cloud in uniform tum; do
for rep in 1 2 3 4; do
  for v in default "$@"; do
    lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
    TC_HIP_LIB=$lib timeout 300 python3 bench.py --cloud $cloud --steps 20 --warmup 4 --no-extras --no-cpu-baseline --no-copy-probe 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cloud $v', 'it/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'icp-only it/s %.0f' % d['icp_only_it_per_s'], 'main pass us %.1f' % d['roofline']['avg_launch_us'], 'normals Mpts/s %.0f' % d['normals_mpts_per_s'])"
  done
done; done
