import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faulthandler, sys, time
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
f = synth.kitti_shaped_cloud(seed=1)
ego = synth.yaw_isometry((-1.0, 0.0, 0.0), -np.deg2rad(0.5))
cur = synth.apply_isometry(ego, f)
cfg = tc.KissIcpConfig(voxel_size=0.5, max_range=100.0, min_range=0.5, max_iterations=50)
print("calling", flush=True)
t0 = time.time()
g = ctx.kiss_icp(cur, f, None, cfg)
print("done", time.time() - t0, g.iterations, g.converged, g.mse, g.transformation, len(g.corr_target), flush=True)
faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(60, exit=True)
init = synth.yaw_isometry((0.9, 0.0, 0.0), np.deg2rad(0.4))
print("init variant", flush=True)
g2 = ctx.kiss_icp(cur, f, init, cfg); print(g2.iterations, g2.converged, g2.mse, flush=True)
import torch
print("device variant", flush=True)
gd = ctx.kiss_icp(torch.from_numpy(cur).cuda(), torch.from_numpy(f).cuda(), None, cfg); print(gd.iterations, flush=True)
for name, fn in (("empty", lambda: ctx.kiss_icp(cur[:0], f, None, cfg)), ("vox0", lambda: ctx.kiss_icp(cur, f, None, tc.KissIcpConfig(voxel_size=0.0))),
                 ("it0", lambda: ctx.kiss_icp(cur, f, None, tc.KissIcpConfig(max_iterations=0))),
                 ("range", lambda: ctx.kiss_icp(cur, f, None, tc.KissIcpConfig(voxel_size=0.5, min_range=500.0, max_range=600.0)))):
    print(name, flush=True)
    try:
        fn(); print("  no error")
    except tc.Error as e:
        print("  ", type(e).__name__, e)
print("oracle", flush=True)
from oracle import oracle as O
r, nd = O.kiss_icp(cur, f, None, 0.5, 100.0, 0.5, 50); print(r.iterations, nd, flush=True)
r2, _ = O.kiss_icp(cur, f, init, 0.5, 100.0, 0.5, 50); print(r2.iterations, flush=True)
