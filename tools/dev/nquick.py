import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: normals kernel time on a few clouds (library events); TC_HIP_LIB selects a variant build"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
cl = [("u1M", synth.uniform_cloud(1_000_000, 2), (16, 10)), ("tum", synth.tum_shaped_cloud(seed=1), (16,)), ("kitti", synth.kitti_shaped_cloud(seed=2), (16,))]
for name, pts, ks in cl:
    d = torch.from_numpy(pts).cuda()
    for k in ks:
        ctx.estimate_normals(d, k)
        ctx.profile_enable(1); ctx.profile_reset()
        for _ in range(3): ctx.estimate_normals(d, k)
        st = ctx.profile_read(); ctx.profile_enable(0)
        print(f"{tag} {name} k={k}: {1e3 * st['normals_knn_pca'][1] / st['normals_knn_pca'][0]:.1f} us", flush=True)
