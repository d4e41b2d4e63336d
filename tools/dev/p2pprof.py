import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, time, threecrate_amd as tc
from threecrate_amd import synth
ctx = tc.GpuContext(0)
pts = synth.tum_shaped_cloud(seed=1, step=2.085)
T = synth.harness_transform()
src = synth.apply_isometry(T, pts)
for dbg in (0,):
    for _ in range(3): ctx.icp_point_to_point(src, pts, None, 10, 1e-5, None)
    ctx.profile_enable(1); ctx.profile_reset()
    t0=time.perf_counter(); r = ctx.icp_point_to_point(src, pts, None, 10, 1e-5, None); dt=time.perf_counter()-t0
    st = ctx.profile_read(); ctx.profile_enable(0)
    print(len(pts), r.iterations, "%.3f ms" % (dt*1e3), {k: (v[0], round(1e3*v[1]/max(v[0],1),1)) for k,v in st.items()})
