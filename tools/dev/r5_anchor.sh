#!/bin/bash
# Run ON THE GPU BOX: the 2-NN anchors of the sorted-trip main pass: parity slice, counters and per-launch traces per margin, bench A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for m in ${MARGINS:-0 0.05 0.1 0.2 0.3}; do
  echo "== margin $m"
  TC_ICP_ANCHOR_MARGIN=$m TC_DEBUG=8 timeout 300 python3 tools/dev/trace.py 2>&1 | grep -E "main pass" | tail -1
  TC_ICP_ANCHOR_MARGIN=$m bash tools/trace_iters.sh m$m | grep -E "mean|^icp|iteration"
done
bash tools/dev/ab_lib.sh "$@"
