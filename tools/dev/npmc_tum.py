import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: the normals call on the TUM-shaped 1 M-point surface (BASELINE configs[2] shape), for rocprofv3 counter passes:
bash tools/pmc_passes.sh ntum tools/dev/npmc_tum.py"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
pts = (synth.tum_shaped_cloud(seed=1) + synth.gaussian_noise(len(synth.tum_shaped_cloud(seed=1)), 200, 1e-3)).astype(np.float32)
dt = torch.from_numpy(pts).cuda()
for i in range(3):
    out = ctx.estimate_normals(dt, 16)
print(float(out[:, 3:].norm(dim=1).mean()))
