#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_trace_ab.sh <variant>... -- tools/trace_iters.sh (per-launch durations of one 50-iteration call) for the default library and each variant
cd "$GRAFT_REPO_ROOT"
for v in default "$@"; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  echo "=== $v"
  TC_HIP_LIB=$lib bash tools/trace_iters.sh ab_$v 2>&1 | grep -E "mean|iteration|last call"
done
