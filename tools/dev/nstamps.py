import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: phase stamps of the normals kernel (-DTC_PHASE_STAMPS build + TC_DEBUG=1024), 1 M points"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1_000_000, 2)).cuda()
for k in (16, 16, 10):
    print("k", k, flush=True)
    ctx.estimate_normals(d, k)
