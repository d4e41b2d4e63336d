"""dev: normals kernel time at 1 M points for a few k (library profile events)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
d = torch.from_numpy(synth.uniform_cloud(1_000_000, seed=1)).cuda()
for k in (16, 10, 32):
    ctx.estimate_normals(d, k)
    ctx.profile_enable(1); ctx.profile_reset()
    for _ in range(5): ctx.estimate_normals(d, k)
    st = ctx.profile_read(); ctx.profile_enable(0)
    print("k", k, {n: round(1e3 * ms / max(c, 1), 1) for n, (c, ms) in st.items() if "normals" in n})
