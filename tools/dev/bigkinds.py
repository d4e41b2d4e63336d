import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: GPU time of the fuzz's calls on the big odd clouds, per cloud kind (no oracle)"""
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
rng = np.random.default_rng(1)
n = 270000
def cloud(kind):
    if kind == 0: p = rng.random((n, 3))
    elif kind == 1: p = rng.random((n, 3)) * np.array([10.0, 3.0, 0.2])
    elif kind == 2: p = np.concatenate([rng.normal(0, 0.05, (n // 2, 3)), rng.normal(3, 0.2, (n - n // 2, 3))])
    elif kind == 3: u = rng.random((n, 2)); p = np.stack([u[:, 0], u[:, 1], 0.1 * np.sin(6 * u[:, 0]) + 1e-4 * rng.normal(size=n)], 1)
    elif kind == 4: t = rng.random(n); p = np.stack([t, 2 * t, -t], 1) + 1e-3 * rng.normal(size=(n, 3))
    elif kind == 5: p = np.round(rng.random((n, 3)) * 8) / 8 + 1e-5 * rng.normal(size=(n, 3))
    else: p = rng.random((n, 3)); p[: n // 4] = p[0]
    return p.astype(np.float32)
def t(fn):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
for kind in range(7):
    for outl in (False, True):
        p = cloud(kind)
        if outl:
            ext = p.max(0) - p.min(0) + 1e-9
            p[5] = p.mean(0) + ext * np.array([50, 0, 0]); p[6] = p[5] + ext * 1e-3; p[7] = p.mean(0) + ext * np.array([0, -80, 30])
        qs = np.concatenate([p[:20], (p.mean(0) + p.std(0) * 4 * rng.normal(size=(10, 3))).astype(np.float32)])
        src = synth.apply_isometry(synth.yaw_isometry((0.01, 0.0, 0.0), 0.01), p[: n // 2])
        a = t(lambda: ctx.find_k_nearest_batch(p, qs, 16))
        b = t(lambda: ctx.icp_detailed(src, p, None, 1, None, 0.0))
        c = t(lambda: ctx.estimate_normals(p, 16))
        d = t(lambda: ctx.voxel_grid_filter(p, float(np.linalg.norm(p.max(0) - p.min(0)) * 0.02)))
        print(f"kind {kind} outliers {outl}: knn {a:8.1f} ms  icp1 {b:8.1f} ms  normals {c:8.1f} ms  voxel {d:7.1f} ms", flush=True)
