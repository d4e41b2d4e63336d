#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/ab_normals.sh <variant>... -- per-kernel times (nprof.py) and wall time (ntime.py) of the 1 M-point
# k = 16 normals call with the default library and each variant, two rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in default "$@"; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  echo "== $v"
  TC_HIP_LIB=$lib timeout 300 python3 tools/dev/nprof.py 2>/dev/null | tail -1
  TC_HIP_LIB=$lib timeout 300 python3 tools/dev/ntime.py 2>/dev/null | tail -1
done; done
