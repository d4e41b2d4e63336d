import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""(TC_DEBUG bits that change the road -- 32, 8192 ... -- need TC_HIP_LIB=threecrate_amd/variants/libthreecrate_hip_dev.so since round 6.)
dev: a few ICP iterations for PMC collection (TC_DEBUG=32 keeps the transform fixed = cold phase)."""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
n = 1000000
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform())
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
nrm = ctx.estimate_normals(dt, 16)
r = ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, 12, None, 0.0, correspondences=False)
print(r.mse, r.iterations)
