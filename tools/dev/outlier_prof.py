import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
ctx.profile_enable(1)
base = synth.uniform_cloud(200000, 1)
for tag, extra in (("1 outlier x30", np.array([[30, 0.5, 0.5]], np.float32)), ("inside far", np.array([[0.5, 0.5, 0.5]], np.float32))):
    pts = np.concatenate([base, extra]).astype(np.float32)
    d = torch.from_numpy(pts).cuda()
    ctx.estimate_normals(d, 16); ctx.profile_reset()
    ctx.estimate_normals(d, 16)
    print(tag, {k: (c, round(ms, 3)) for k, (c, ms) in ctx.profile_read().items()})
    q = torch.from_numpy(extra).cuda()
    ctx.profile_reset()
    ctx.find_k_nearest_batch(pts, extra, 17)
    print("  knn of that point alone:", {k: (c, round(ms, 3)) for k, (c, ms) in ctx.profile_read().items()})
