import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
n = 1000000
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform())
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
for prof in [False, True, False, True]:
    ctx.profile_enable(prof)
    for rep in range(3):
        t0 = time.perf_counter(); nrm = ctx.estimate_normals(dt, 16); t1 = time.perf_counter()
        r = ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, 50, None, 0.0, correspondences=False); t2 = time.perf_counter()
    print(f"profiling={prof}: normals {1e3*(t1-t0):.3f} ms  icp50 {1e3*(t2-t1):.3f} ms")
    if prof: ctx.profile_reset()
