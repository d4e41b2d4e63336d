#!/bin/bash
# Run ON THE GPU BOX: bench.py's step with the binned index placement (default) and with the atomic counting sort only
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for mode in 1 0; do
    TC_INDEX_BINNED=$mode python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-copy-probe 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('TC_INDEX_BINNED=$mode', 'it/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'icp-only it/s %.0f' % d['icp_only_it_per_s'], 'normals Mpts/s %.0f' % d['normals_mpts_per_s'])"
  done
done
