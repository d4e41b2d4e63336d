import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: an ICP call that converges at once (huge threshold): the rest of its first chunk are early-exit launches = the fixed
cost of each kernel of an iteration (dispatch + the `done` check), under rocprofv3 --kernel-trace."""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
n = 1000000
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
nrm = ctx.estimate_normals(dt, 16)
for rep in range(3):
    r = ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, 50, None, 1e9, correspondences=False)
print(r.mse, r.iterations, r.converged)
