import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: per-kernel times of a normals call on a cloud with far outliers"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
for n in (200000, 1000000):
    base = synth.uniform_cloud(n, 1)
    for name, extra in (("clean", np.zeros((0, 3), np.float32)), ("1 outlier x30", np.array([[30, 0.5, 0.5]], np.float32)),
                        ("3 outliers x100", np.array([[100, 0.5, 0.5], [0.5, -100, 0.2], [0.3, 0.3, 100]], np.float32))):
        d = torch.from_numpy(np.concatenate([base, extra]).astype(np.float32)).cuda()
        ctx.estimate_normals(d, 16)
        ctx.profile_enable(1); ctx.profile_reset()
        ctx.estimate_normals(d, 16)
        st = ctx.profile_read(); ctx.profile_enable(0)
        print(n, name, {k: round(1e3 * v[1] / max(v[0], 1), 1) for k, v in st.items()}, flush=True)
