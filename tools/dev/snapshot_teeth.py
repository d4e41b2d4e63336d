import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
tc.GpuContext._release = lambda self, d: None          # no ordering at all
pts = synth.uniform_cloud(400_000, seed=5)
rh = tc.Cloud(ctx, pts); ref = rh.estimate_normals(10); rh.close()
bad = 0
for _ in range(20):
    x = torch.from_numpy(pts).cuda()
    h = tc.Cloud(ctx, x)
    x.zero_()
    got = h.estimate_normals(10); h.close()
    bad += int(not np.array_equal(got.cpu().numpy(), ref))
print("handles that saw the overwritten tensor without the stream wait:", bad, "of 20")
