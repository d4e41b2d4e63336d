#!/bin/bash
# Run ON THE GPU BOX: the round's differential campaign on the final build -- every fuzzer for <seconds> (default 240) with a
# fresh seed, summaries to gpurun_out/campaign_<seed>.txt
cd "$GRAFT_REPO_ROOT"
SEC=${1:-240}; SEED=${2:-401}
OUT=gpurun_out/campaign_$SEED.txt
: > $OUT
for f in normals_fuzz fuzz loop_fuzz variants_fuzz vor_fuzz; do
  echo "== $f.py $SEC $SEED" | tee -a $OUT
  timeout $((SEC + 120)) python3 tools/dev/$f.py $SEC $SEED 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $OUT
done
