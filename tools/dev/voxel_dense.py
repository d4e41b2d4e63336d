import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: voxel filter with ~1000 points per voxel (TUM-shaped frame, 0.2 m voxels) under rocprofv3 --stats"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
pts = synth.tum_shaped_cloud(seed=1, step=2.085)
pts = (pts + np.random.default_rng(0).normal(0, 1e-4, pts.shape)).astype(np.float32)
for _ in range(5):
    out = ctx.voxel_grid_filter(pts, 0.2)
print(len(pts), len(out))
