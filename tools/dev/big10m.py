"""per-kernel times of the 10 M-point configuration (configs[3]) on one GPU"""
import sys, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth, distributed as D
ctx = tc.GpuContext(0)
n = 10_000_000
for noise in (0.0, 1e-3):
    src, tgt, T = synth.registration_pair(n, seed=7, scale=(10.0, 10.0, 1.0), noise_sigma=noise)
    ds, dt = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    comm = D.Comm.local(ctx)
    nrm = D.sharded_estimate_normals(ctx, dt, 16, comm=comm)
    D.sharded_icp_point_to_plane(ctx, ds, dt, nrm, None, 50, None, 0.0, comm=comm)
    ctx.profile_enable(1); ctx.profile_reset()
    r = D.sharded_icp_point_to_plane(ctx, ds, dt, nrm, None, 50, None, 0.0, comm=comm)
    st = ctx.profile_read(); ctx.profile_enable(0)
    print("noise", noise, "mse", r.mse, {k: (c, round(1e3 * ms / max(c, 1), 1)) for k, (c, ms) in st.items()})
