import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: host-buffer entry points (PCIe inclusive) vs device-resident, 1M points."""
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
n = 1000000
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform())
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
for rep in range(3):
    t0 = time.perf_counter(); nh = ctx.estimate_normals(tgt, 16); t1 = time.perf_counter()
    rh = ctx.icp_point_to_plane_detailed(src, tgt, nh, None, 50, None, 0.0); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter(); nd = ctx.estimate_normals(dt, 16); torch.cuda.synchronize(); t4 = time.perf_counter()
    rd = ctx.icp_point_to_plane_detailed(ds, dt, nd, None, 50, None, 0.0, correspondences=False); t5 = time.perf_counter()
print(f"host buffers  : normals {1e3*(t1-t0):.2f} ms, icp50 (with correspondences out) {1e3*(t2-t1):.2f} ms")
print(f"device buffers: normals {1e3*(t4-t3):.2f} ms, icp50 {1e3*(t5-t4):.2f} ms")
