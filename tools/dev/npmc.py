import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
dt = torch.from_numpy(synth.uniform_cloud(1000000, 1)).cuda()
for i in range(3):
    out = ctx.estimate_normals(dt, 16)
print(float(out[:,3:].norm(dim=1).mean()))
