#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_dense.sh <variant> -- a variant of the ICP main pass: parity tests WITH the variant, then bench A/B, per-launch trace, TUM pair
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1
LIB="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$V.so"
mkdir -p gpurun_out/r6
TC_HIP_LIB=$LIB timeout 1500 python3 -m pytest tests -m gpu -x -q -k "(icp or golden or kats or pipeline or sharded or loop or stream or cloud or fuzz) and not debug_bits and not abi" 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -4 | tee gpurun_out/r6/pytest_$V.txt
# the second-neighbour certificate on against off (TC_DEBUG=4096: needs the variant with the development api object, <variant>dev), bit for bit
if [ -f "$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_${V}dev.so" ]; then
  for seed in 1 2 3; do TC_HIP_LIB="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_${V}dev.so" timeout 600 python3 tools/dev/vor_fuzz.py 60 $seed second 2>&1 | tail -2; done | tee gpurun_out/r6/vorfuzz_$V.txt
fi
shift
bash tools/dev/ab_lib.sh $V "$@" 2>&1 | tee gpurun_out/r6/ab_$V.txt
bash tools/dev/r6_trace_ab.sh $V "$@" 2>&1 | tee gpurun_out/r6/trace_$V.txt
for v in default $V "$@"; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  TC_HIP_LIB=$lib timeout 300 python3 bench.py --cloud tum --steps 6 --warmup 2 --no-cpu-baseline --no-copy-probe --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tum $v', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'main pass us %.1f' % d['roofline']['avg_launch_us'])" | tee -a gpurun_out/r6/ab_$V.txt
done
