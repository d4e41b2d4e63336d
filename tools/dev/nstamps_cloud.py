import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: phase stamps of the normals kernel (-DTC_PHASE_STAMPS build + TC_DEBUG=1024) on a named cloud: uniform | tum | kitti | sheet"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "tum"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
pts = {"uniform": lambda: synth.uniform_cloud(1_000_000, 2), "tum": lambda: synth.tum_shaped_cloud(seed=1), "kitti": lambda: synth.kitti_shaped_cloud(seed=2),
       "sheet": lambda: (np.random.default_rng(3).normal(0, 1, (200000, 3)) * np.array([1, 1, 0.02])).astype(np.float32)}[name]()
ctx = tc.GpuContext(0)
d = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
print(name, len(pts), "k", k, flush=True)
ctx.estimate_normals(d, k)
ctx.estimate_normals(d, k)
