#!/bin/bash
# dev: build a variant of the library with extra compiler flags (A/B experiments on the GPU box):
#   bash tools/dev/build_variant.sh <name> "<extra flags>" [files...]  ->  threecrate_amd/variants/libthreecrate_hip_<name>.so
# Only the listed source files (default: icp) are recompiled with the flags; the rest are the objects of the default build
# (run `make -C threecrate_amd/csrc` first).  Git-ignored, travels with gpurun; select with TC_HIP_LIB=<path>.
set -eu
NAME=$1; EXTRA=${2:-}; shift; shift || true
FILES=${@:-icp}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/threecrate_amd/csrc
OBJ=$ROOT/build/var_$NAME
rm -rf "$OBJ"; mkdir -p "$OBJ" "$ROOT/threecrate_amd/variants"
for f in api grid normals icp voxel stream comm cloud; do cp "$SRC/$f.o" "$OBJ/$f.o"; done
pids=()
for spec in $FILES; do
  # "icp" compiles csrc/icp.hip; "icp=/some/other/icp.hip" compiles that file in its place (e.g. an older revision: git show REV:path > file)
  f=${spec%%=*}; src="$SRC/$f.hip"; [ "$spec" != "$f" ] && src=${spec#*=}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wall -Wno-unused-result -I"$SRC" $EXTRA -c "$src" -o "$OBJ/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/threecrate_amd/variants/libthreecrate_hip_$NAME.so" "$OBJ"/*.o -lpthread -ldl
echo "built threecrate_amd/variants/libthreecrate_hip_$NAME.so"
