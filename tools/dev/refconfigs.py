import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: the reference's own published task shapes (docs/benchmarks.md: normals k=10; icp_point_to_point <= 10 iterations,
threshold 1e-5, target = source moved by (0.05, -0.02, 0.01) + 0.02 rad yaw; voxel 0.2 m) on synthetic clouds of the
same sizes and shapes.  Host buffers in, host buffers out (the reference's call shape), median of 5 after 2 warm-ups."""
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
T = synth.harness_transform()
def med(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))
for name, pts in (("TUM-shaped ~230k", synth.tum_shaped_cloud(seed=1, step=2.085)), ("KITTI-shaped 120k", synth.kitti_shaped_cloud(seed=1)),
                  ("uniform 35k", synth.uniform_cloud(35000, 1, (40.0, 40.0, 3.0)))):
    rng = np.random.default_rng(0)
    pts = (pts + rng.normal(0, 1e-4, pts.shape)).astype(np.float32)
    tgt = synth.apply_isometry(T, pts)
    n = len(pts)
    tn = med(lambda: ctx.estimate_normals(pts, 10))
    r = [None]
    def icp():
        r[0] = ctx.icp_point_to_point(pts, tgt, None, 10, 1e-5, None, correspondences=False)
    ti = med(icp)
    tv = med(lambda: ctx.voxel_grid_filter(pts, 0.2 if name.startswith("KITTI") or name.startswith("uniform") else 0.02))
    print(f"{name:18s} n={n:7d}  normals k=10 {tn:7.3f} ms ({n/tn/1e3:7.1f} Mpts/s)   icp_p2p<=10 {ti:7.3f} ms ({r[0].iterations} it)   voxel {tv:6.3f} ms")
