import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: search statistics of the TUM-shaped pair by iteration range (counting instantiation, calls of k iterations differenced)"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
base = synth.tum_shaped_cloud(seed=1); n = len(base)
src = (synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), base) + synth.gaussian_noise(n, 100, 1e-3)).astype(np.float32)
tgt = (base + synth.gaussian_noise(n, 200, 1e-3)).astype(np.float32)
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
t = tc.Cloud(ctx, dt); t.estimate_normals(16, out=False); s = tc.Cloud(ctx, ds)
prev = None
for k in (8, 12, 20, 35, 50):
    ctx.profile_enable(3)
    s.icp_point_to_plane(t, None, k, None, 0.0)
    st = ctx.search_stats(); ctx.profile_enable(0)
    cur = {kk: st[kk] for kk in ("wave_trips", "wave_trips_without_a_search", "searches", "candidate_steps_needed", "candidate_steps_taken_by_slowest_lanes")}
    if prev is not None:
        d = {kk: cur[kk] - prev[1][kk] for kk in cur}; its = k - prev[0]
        print(f"iterations {prev[0] + 1}..{k}: searches per point-iteration {d['searches'] / (n * its):.4f}  trips without a search {d['wave_trips_without_a_search'] / max(d['wave_trips'], 1):.3f}  "
              f"steps per searching trip (slowest lane) {d['candidate_steps_taken_by_slowest_lanes'] / max(d['wave_trips'] - d['wave_trips_without_a_search'], 1):.1f}  steps per search {d['candidate_steps_needed'] / max(d['searches'], 1):.1f}")
    prev = (k, cur)
