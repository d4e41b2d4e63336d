#!/bin/bash
# Run ON THE GPU BOX: first look at the sorted-trip main pass (round 5): parity tests, lock-step counters, A/B against r04 and the unsorted deal
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5
for v in default nosort; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  echo "== lock-step counters: $v"
  TC_HIP_LIB=$lib TC_DEBUG=8 timeout 300 python3 tools/dev/trace.py 2>&1 | grep -E "main pass|^[0-9]" | tail -3
done
bash tools/dev/ab_lib.sh r04 nosort
bash tools/dev/ab_iters.sh base r04
