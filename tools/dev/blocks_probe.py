import sys; sys.path.insert(0, ".")
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
T = synth.harness_transform()
src, tgt, _ = synth.registration_pair(1_000_000, seed=1, transform=T, noise_sigma=1e-4)
ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
t = tc.Cloud(ctx, dt); t.estimate_normals(16, out=False); s = tc.Cloud(ctx, ds)
for iters in (5, 5, 20, 50):
    print("== iterations", iters, flush=True)
    s.icp_point_to_plane(t, None, iters, None, 0.0)
