import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth
from tests.helpers import cos_abs
ctx = tc.GpuContext(0)
n = 60000
rng = np.random.default_rng(5)
base = synth.uniform_cloud(n, 31, (4.0, 3.0, 1.0))
out = np.array([[120, 1.5, 0.5], [120.004, 1.5, 0.5], [2, -300, 0.4], [1, 2, 90], [-50, -60, -70]], np.float32)
for tag, pts in (("clean", base), ("outliers", None)):
    if pts is None:
        pts = base.copy(); where = rng.integers(0, n, len(out)); pts[where] = out
    g = ctx.estimate_normals(pts, 10); r = O.estimate_normals(pts, 10)
    c = cos_abs(g[:, 3:6], r[:, 3:6])
    bad = np.nonzero(c < 1 - 1e-4)[0]
    print(tag, "bad", len(bad), bad[:10], c[bad[:10]])
    if len(bad):
        i = bad[0]
        gi, gd, gc = ctx.find_k_nearest_batch(pts, pts[i:i+1], 11)
        oi, od, oc = O.knn_batch(pts, pts[i:i+1], 11)
        print(" point", pts[i], "knn equal", np.array_equal(gd, od), gd[0][-3:], "normal g", g[i, 3:], "r", r[i, 3:])
