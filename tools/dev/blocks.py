"""TC_DEBUG=1024: schedule of the blocks of the last main pass of an ICP call (moving phase: 10 iterations; aligned: from the answer)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(1_000_000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
nrm = ctx.estimate_normals(dt, 16)
for iters, init in ((1, None), (10, None), (20, None), (12, synth.harness_transform())):
    print("iterations", iters, "from", "identity" if init is None else "the answer", flush=True)
    for _ in range(2):
        ctx.icp_point_to_plane_detailed(ds, dt, nrm, init, iters, None, 0.0, correspondences=False)
