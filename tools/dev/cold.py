import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""(TC_DEBUG bits that change the road -- 32, 8192 ... -- need TC_HIP_LIB=threecrate_amd/variants/libthreecrate_hip_dev.so since round 6.)
dev: cold-phase kernel time (TC_DEBUG=32 keeps the transform fixed) for different source offsets."""
import os, sys, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
n = 1000000
ctx = tc.GpuContext(0)
ctx.profile_enable(True)
for name, T in [("identity-ish small", synth.small_transform(n)), ("harness", synth.harness_transform()),
                ("half-cell shift", synth.yaw_isometry((0.005, 0.0, 0.0), 0.0)), ("2-cell shift x", synth.yaw_isometry((0.0226, 0.0, 0.0), 0.0))]:
    src, tgt, _ = synth.registration_pair(n, seed=1, transform=T)
    dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
    nrm = ctx.estimate_normals(dt, 16)
    ctx.profile_reset()
    try:
        r = ctx.icp_point_to_plane_detailed(ds, dt, nrm, None, 10, None, 0.0, correspondences=False)
    except Exception as e:
        print("err", e)
    st = ctx.profile_read()
    c, ms = st["icp_correspond_reduce_p2plane"]
    print(f"{name:22s} {1e3*ms/c:8.1f} us avg over {c}")
