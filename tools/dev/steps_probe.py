import sys, time; sys.path.insert(0, ".")
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
T = synth.harness_transform()
src, tgt, _ = synth.registration_pair(1_000_000, seed=1, transform=T, noise_sigma=1e-4)
ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
ctx.profile_enable(2)
ts = []
for i in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = tc.Cloud(ctx, dt); n = t.estimate_normals(16); s = tc.Cloud(ctx, ds)
    r = s.icp_point_to_plane(t, None, 50, None, 0.0, correspondences="device")
    t.close(); s.close()
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("ms per step:", " ".join(f"{x:.3f}" for x in ts))
