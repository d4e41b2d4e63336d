import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
ctx.profile_enable(1)
for name, pts in [("tum 1M surface", synth.tum_shaped_cloud(seed=1)), ("kitti 120k", synth.kitti_shaped_cloud(seed=1)), ("uniform 1M", synth.uniform_cloud(1000000, 1))]:
    rng = np.random.default_rng(0)
    pts = (pts + rng.normal(0, 1e-4, pts.shape)).astype(np.float32)
    d = torch.from_numpy(pts).cuda()
    src = torch.from_numpy(synth.apply_isometry(synth.yaw_isometry((-0.01, 0.004, 0.002), -0.002), pts)).cuda()
    for rep in range(2):
        ctx.profile_reset()
        t0 = time.perf_counter(); nrm = ctx.estimate_normals(d, 16); t1 = time.perf_counter()
        r = ctx.icp_point_to_plane_detailed(src, d, nrm, None, 30, None, 0.0, correspondences=False); t2 = time.perf_counter()
    st = ctx.profile_read()
    print(f"{name}: n={len(pts)} normals {1e3*(t1-t0):.2f} ms, icp30 {1e3*(t2-t1):.2f} ms, mse {r.mse:.2e}")
    print("   ", {k: round(1e3*ms/max(c,1),1) for k,(c,ms) in st.items() if k in ("normals_knn_pca","icp_correspond_reduce_p2plane","icp_refine","cell_rank_gather","cell_hist")})
