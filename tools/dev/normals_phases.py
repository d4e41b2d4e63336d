"""TC_DEBUG=1024 with a -DTC_PHASE_STAMPS build: where a wave of the normals kernel spends its time (1 M points, k = 16)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
pts = synth.uniform_cloud(1_000_000, seed=1)
d = torch.from_numpy(pts).cuda()
for k in (16, 16, 10, 32):
    print("k", k, flush=True)
    ctx.estimate_normals(d, k)
