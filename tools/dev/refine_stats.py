"""dev: refine-pass statistics (TC_DEBUG=64; exit-ring histogram with a -DTC_REFINE_STATS build) per number of iterations"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(1_000_000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
nrm = ctx.estimate_normals(dt, 16)
for iters in (1, 2, 5, 10, 20, 30, 40, 50):
    print("iterations", iters, flush=True)
    ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, iters, None, 0.0, correspondences=False)
