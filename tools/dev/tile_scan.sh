#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/tile_scan.sh "8,5,4" "16,4,3" ... -- the bench (no extras) per source-tile shape (TC_ICP_TILE)
cd "$GRAFT_REPO_ROOT"
for t in "$@"; do
  TC_ICP_TILE=$t timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-copy-probe --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile $t', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'main us %.1f' % d['roofline']['avg_launch_us'])"
done
