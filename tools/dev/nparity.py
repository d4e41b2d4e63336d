import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: how close are the HIP normals to the oracle's (bit level) on uniform clouds"""
import numpy as np, torch, threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth
ctx = tc.GpuContext(0)
for n, k, orient in ((60000, 10, True), (200000, 16, True), (50000, 20, False)):
    pts = synth.uniform_cloud(n, 31, (4.0, 3.0, 1.0))
    cfg = tc.NormalEstimationConfig(k_neighbors=k, consistent_orientation=orient)
    g = ctx.estimate_normals_with_config(pts, cfg)
    r = O.estimate_normals(pts, k, None, orient)
    a, b = g[:, 3:6].astype(np.float64), r[:, 3:6].astype(np.float64)
    c = np.abs((a * b).sum(1))
    print(f"n={n} k={k} orient={orient}: bit-identical {np.all(g == r, axis=1).mean():.6f}, worst 1-|cos| {1 - c.min():.3e}, beyond 1e-4: {(c < 1 - 1e-4).sum()}, signs equal {(np.sign((a*b).sum(1)) > 0).mean():.6f}")
