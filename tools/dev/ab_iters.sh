#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/ab_iters.sh <name> [<name> ...] -- per-launch ICP kernel durations (tools/trace_iters.sh)
# with the default library ("base") and with each named variant of tools/dev/build_variant.sh
set -u
for v in "$@"; do
  if [ "$v" = base ]; then unset TC_HIP_LIB; else export TC_HIP_LIB=$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so; fi
  echo "== $v"
  bash tools/trace_iters.sh ab_$v
done
