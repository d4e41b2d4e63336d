#!/bin/bash
# Run ON THE GPU BOX: bash tools/dev/r6_tail.sh <variant>... -- the iteration tail by phase (bench.py's phase_kernels_us: main / refine / finalize in the
# moving and in the converged phase) for the default library and the variants, two rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for v in default "$@"; do
    lib=""; [ "$v" != default ] && lib="$GRAFT_REPO_ROOT/threecrate_amd/variants/libthreecrate_hip_$v.so"
    TC_HIP_LIB=$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-copy-probe 2>/dev/null | tail -1 | \
      python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ph=d.get('phase_kernels_us',{})
print('$v', 'it/s %.0f' % d['value'], 'icp-only %.0f' % d['icp_only_it_per_s'], 'iteration us %.2f' % d['roofline']['iteration']['us'], {p:{k.replace('icp_',''):v for k,v in ph[p].items() if k in ('icp_correspond_reduce_p2plane','icp_refine','icp_finalize')} for p in ph})"
  done
done
