#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/ab_normals_tum.sh <variant>... -- normals kernel time on the TUM-shaped 1 M-point surface and on the uniform
# cloud (bench.py --cloud tum / default, no extras) with the default library and each variant, two rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in default "$@"; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  for cloud in tum uniform; do
  TC_HIP_LIB=$lib timeout 300 python3 bench.py --cloud $cloud --steps 6 --warmup 2 --no-cpu-baseline --no-copy-probe 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$cloud', 'it/s %.0f' % d['value'], 'normals Mpts/s %.0f' % d['normals_mpts_per_s'], 'normals kernel us', d.get('normals_kernels_us',{}).get('normals_knn_pca'))"
  done
done; done
