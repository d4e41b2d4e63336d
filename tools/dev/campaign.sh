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
# round 6: the second-neighbour certificate + dense trips against the development build with TC_DEBUG=4096 (certificate off)
echo "== vor_fuzz.py $SEC $SEED second" | tee -a $OUT
timeout $((2 * SEC + 240)) python3 tools/dev/vor_fuzz.py $SEC $SEED second 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $OUT
for f in paths_stress index_stress; do
  echo "== $f.py $SEC $SEED" | tee -a $OUT
  timeout $((SEC + 120)) python3 tools/dev/$f.py $SEC $SEED 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT
done
