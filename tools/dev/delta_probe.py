import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: how far the k-th update of a registration moves the far corner of the target's box, against the cell edge (the switch of the
second-neighbour certificate: icp.hip compose()): python tools/dev/delta_probe.py [tum|uniform]"""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
from oracle import oracle as O
which = sys.argv[1] if len(sys.argv) > 1 else "tum"
ctx = tc.GpuContext(0)
if which == "tum":
    base = synth.tum_shaped_cloud(seed=1); n = len(base)          # the pair of bench.py measure_tum_pair
    src = (synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -np.deg2rad(0.3)), base) + synth.gaussian_noise(n, 100, 1e-3)).astype(np.float32)
    tgt = (base + synth.gaussian_noise(n, 200, 1e-3)).astype(np.float32)
else:
    src, tgt, _ = synth.registration_pair(1_000_000, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
t = tc.Cloud(ctx, dt); t.estimate_normals(16, out=False); s = tc.Cloud(ctx, ds)
far = float(np.linalg.norm(np.maximum(np.abs(tgt.min(0)), np.abs(tgt.max(0)))))
prev = None
for k in range(1, 51):
    r = s.icp_point_to_plane(t, None, k, None, 0.0)
    M = O.isometry_to_matrix(np.asarray(r.transformation, np.float32)).astype(np.float64)
    if prev is not None:
        D = M @ np.linalg.inv(prev)
        ang = float(np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1))); tn = float(np.linalg.norm(D[:3, 3]))
        if k <= 12 or k % 5 == 0: print(f"update {k:2d}: angle {ang:.3e} |t| {tn:.3e}  moves the far corner (|x| = {far:.2f}) by <= {ang * far + tn:.3e}   mse {r.mse:.6e}")
    prev = M
