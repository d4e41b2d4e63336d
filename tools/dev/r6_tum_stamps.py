import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev (round 6): TC_DEBUG=1024 with a -DTC_PHASE_STAMPS build of icp.hip on the TUM-shaped pair: where a wave of the LAST main pass of a
50-iteration call (a certified pass) spends its time, and the block schedule.  Two calls: the second starts with the context's hint."""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
base = synth.tum_shaped_cloud(seed=1); n = len(base)
src = (synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), base) + synth.gaussian_noise(n, 100, 1e-3)).astype(np.float32)
tgt = (base + synth.gaussian_noise(n, 200, 1e-3)).astype(np.float32)
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
t = tc.Cloud(ctx, dt); t.estimate_normals(16, out=False)
for rep in range(3):
    s = tc.Cloud(ctx, ds)
    print("call", rep, flush=True)
    r = s.icp_point_to_plane(t, None, 50, None, 0.0)
    s.close()
print(r.mse, r.iterations)
