#!/bin/bash
# Run ON THE GPU BOX: bench.py --mode sharded (10 M points, one rank) with the default library and variant builds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for v in default "$@"; do
    lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
    TC_HIP_LIB=$lib timeout 300 python3 bench.py --mode sharded --steps 3 --warmup 1 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sharded_10m $v', 'it/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'])"
  done
done
