import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: hipEvent kernel times of one estimate_normals(k=16) call on the 1 M-point bench cloud"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1000000, 2)).cuda()
ctx.profile_enable(1)
for _ in range(3): ctx.estimate_normals(d, 16)
ctx.profile_reset()
for _ in range(10): ctx.estimate_normals(d, 16)
st = ctx.profile_read()
print({k: round(1e3 * ms / max(c, 1), 1) for k, (c, ms) in st.items() if c})
