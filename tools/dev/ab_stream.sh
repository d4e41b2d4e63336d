#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/ab_stream.sh <variant>... -- the frame-stream extra (bench.py --mode stream) with the default library
# and each variant, three rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for v in default "$@"; do
  lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
  TC_HIP_LIB=$lib timeout 300 python3 bench.py --mode stream --steps 50 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'frames/s %.0f' % d['value'], 'us/frame %.1f' % (1e3*d['ms_per_step']))"
done; done
