import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: per-kernel profile of one KITTI-shaped frame pipeline (voxel + normals + icp)."""
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
pts = synth.kitti_shaped_cloud(seed=1)
d = torch.from_numpy(pts).cuda()
src = torch.from_numpy(synth.apply_isometry(synth.yaw_isometry((-0.05, 0.02, 0.01), -0.004), pts)).cuda()
for mode in (0, 1):
    ctx.profile_enable(mode)
    for rep in range(3):
        ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tv = ctx.voxel_grid_filter(d, 0.2); sv = ctx.voxel_grid_filter(src, 0.2)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        nrm = ctx.estimate_normals(tv, 16)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        r = ctx.icp_point_to_plane_detailed(sv, tv, nrm, None, 50, None, 1e-6, correspondences=False)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"mode {mode}: voxel {1e3*(t1-t0):.3f} ms ({len(tv)} pts)  normals {1e3*(t2-t1):.3f} ms  icp {1e3*(t3-t2):.3f} ms ({r.iterations} it)")
    if mode:
        for k, (c, ms) in sorted(ctx.profile_read().items(), key=lambda kv: -kv[1][1]):
            print(f"    {k:34s} {c:4d} x {1e3*ms/max(c,1):8.2f} us = {ms:7.3f} ms")
