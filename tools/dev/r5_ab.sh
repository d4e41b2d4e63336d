#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r5_ab.sh <variant>... -- ICP parity slice, bench A/B (default vs variants), per-launch traces
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "icp or sharded or kiss or gicp" 2>&1 | tail -3
bash tools/dev/ab_lib.sh "$@"
bash tools/dev/ab_iters.sh base "$@"
