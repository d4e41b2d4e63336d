import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import json, numpy as np, threecrate_amd as tc
from oracle import oracle as O
from tests.helpers import sphere_cloud
ctx = tc.GpuContext(0)
s, nn = sphere_cloud(100)
t = s + np.array([0.15, 0, 0], np.float32)
for it in range(1, 12):
    g = ctx.icp_point_to_plane_detailed(s, t, nn, None, it, None, 0.0); o = O.icp_point_to_plane_detailed(s, t, nn, None, it, None, 0.0)
    same = np.array_equal(g.correspondences, o.correspondences)
    print(it, g.mse, o.mse, same, g.transformation[4:], o.transformation[4:])
g = ctx.icp_point_to_plane(s, t, nn, None, 50); o = O.icp_point_to_plane(s, t, nn, None, 50)
print(g.iterations, g.converged, g.mse, o.iterations, o.converged, o.mse)
