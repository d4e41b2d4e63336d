import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: what a few far outliers do to the dense grid (bbox stretched)"""
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
def med(fn, reps=3):
    fn(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
for n in (200000,) if len(sys.argv) > 1 else (200000, 1000000):
    base = synth.uniform_cloud(n, 1)
    for tag, extra in (("clean", np.zeros((0, 3), np.float32)), ("1 outlier x30", np.array([[30, 0.5, 0.5]], np.float32)),
                       ("3 outliers x100", np.array([[100, 0.5, 0.5], [0.5, -100, 0.2], [0.3, 0.3, 100]], np.float32))):
        pts = np.concatenate([base, extra]).astype(np.float32)
        d = torch.from_numpy(pts).cuda()
        tn = med(lambda: ctx.estimate_normals(d, 16))
        src = torch.from_numpy(synth.apply_isometry(synth.invert_isometry(synth.harness_transform())[:3] if False else synth.yaw_isometry((-0.05, 0.02, -0.01), -0.02), pts)).cuda()
        nrm = ctx.estimate_normals(d, 16)
        ti = med(lambda: ctx.icp_point_to_plane_detailed(src, d, nrm, None, 10, None, 0.0, correspondences=False))
        print(f"n={n:8d} {tag:16s} normals {tn:9.2f} ms   icp 10 it {ti:9.2f} ms", flush=True)
