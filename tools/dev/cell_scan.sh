#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/cell_scan.sh <factor>... -- bench (no extras) per TC_ICP_CELL_FACTOR (target cell edge of the ICP
# index in units of the mean spacing; default 1.13), two rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for f in "$@"; do
  TC_ICP_CELL_FACTOR=$f timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-copy-probe --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels_us_avg',{}); print('factor $f', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'main us %.1f' % d['roofline']['avg_launch_us'], 'iteration us %.2f' % d['roofline']['iteration']['us'])"
done; done
