#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_cert.sh <baseline variant> -- the default library (dense trips + second-neighbour certificate in their own
# instantiation of the main pass) against a baseline build: parity tests, the certificate's fuzz A/B, bench A/B (uniform + TUM-shaped), per-launch trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$1
mkdir -p gpurun_out/r6
timeout 1700 python3 -m pytest tests -m gpu -x -q -k "(icp or golden or kats or pipeline or sharded or loop or stream or cloud or fuzz or debug_bits)" 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -4 | tee gpurun_out/r6/pytest_cert.txt
for seed in 1 2 3; do timeout 600 python3 tools/dev/vor_fuzz.py 60 $seed second 2>&1 | tail -2; done | tee gpurun_out/r6/vorfuzz_cert.txt
bash tools/dev/ab_lib.sh $B 2>&1 | tee gpurun_out/r6/ab_cert.txt
bash tools/dev/r6_trace_ab.sh $B 2>&1 | tee gpurun_out/r6/trace_cert.txt
for rep in 1 2; do
for v in default $B; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  TC_HIP_LIB=$lib timeout 300 python3 bench.py --cloud tum --steps 6 --warmup 2 --no-cpu-baseline --no-copy-probe --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tum $v', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'main pass us %.1f' % d['roofline']['avg_launch_us'])" | tee -a gpurun_out/r6/ab_cert.txt
done; done
