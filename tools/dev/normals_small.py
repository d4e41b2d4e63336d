"""dev: normals kernel time on small clouds (a voxel-filtered LiDAR frame: 25 k points) and at 1 M points"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
frame = ctx.voxel_grid_filter(synth.kitti_shaped_cloud(seed=1), 0.2)
for name, pts in (("lidar 25k", frame), ("uniform 100k", synth.uniform_cloud(100_000, seed=1)), ("uniform 1M", synth.uniform_cloud(1_000_000, seed=1))):
    d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    ctx.estimate_normals(d, 16)
    ctx.profile_enable(1); ctx.profile_reset()
    for _ in range(5): ctx.estimate_normals(d, 16)
    st = ctx.profile_read(); ctx.profile_enable(0)
    print(name, len(pts), {n: round(1e3 * ms / max(c, 1), 1) for n, (c, ms) in st.items() if "normals" in n})
