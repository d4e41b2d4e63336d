#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_icp.sh <tag> <variant>... -- ICP parity tests first, then the bench against the variants (three rounds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
mkdir -p gpurun_out/r6
timeout 1700 python3 -m pytest tests -m gpu -x -q -k "icp or golden or kats or pipeline or sharded or loop or stream or cloud or debug_bits" 2>&1 | tail -6 | tee gpurun_out/r6/pytest_icp_$TAG.txt
bash tools/dev/ab_lib.sh "$@" 2>&1 | tee gpurun_out/r6/ab_icp_$TAG.txt
