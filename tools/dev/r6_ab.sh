#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_ab.sh <tag> <variant>... -- normals kernel A/B (default library + variants, two rounds), lock-step
# statistics of the default source (variant nstats, if built), then the normals parity tests and the normals fuzz slice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
mkdir -p gpurun_out/r6
bash tools/dev/ab_normals.sh "$@" 2>&1 | grep -A1 "^==" | grep -v "^--" | tee gpurun_out/r6/ab_$TAG.txt
if [ -f threecrate_amd/variants/libthreecrate_hip_nstats.so ]; then
  TC_HIP_LIB="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_nstats.so" timeout 300 python3 tools/dev/npmc.py 2>&1 | grep "\[tc\]" | tail -2 | tee gpurun_out/r6/nstats_$TAG.txt
fi
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "normals or golden or kats" 2>&1 | tail -5 | tee gpurun_out/r6/pytest_$TAG.txt
