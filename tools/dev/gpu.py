import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev script: quick timing of the hot path on the GPU box (not part of the product)."""
import sys, time
import numpy as np
import torch
import threecrate_amd as tc
from threecrate_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform())
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
torch.cuda.synchronize()
ctx.profile_enable(True)
for rep in range(3):
    ctx.profile_reset()
    t0 = time.time(); nrm = ctx.estimate_normals(dt, 16); t1 = time.time()
    try:
        r = ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, 50, None, 0.0, correspondences=False)
    except Exception as e:
        class R: mse=-1; iterations=-1
        r = R()
    t2 = time.time()
    print(f"rep {rep}: normals {1e3*(t1-t0):.2f} ms ({n/(t1-t0)/1e6:.1f} Mpts/s)  icp50 {1e3*(t2-t1):.2f} ms ({50/(t2-t1):.1f} it/s) mse {r.mse:.3e} it {r.iterations}")
    for k, (cnt, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1]):
        print(f"    {k:36s} {cnt:5d} launches {ms:9.3f} ms total {1e3*ms/max(cnt,1):9.2f} us avg")
