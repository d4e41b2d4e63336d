import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: 2-NN anchor statistic (build: tools/dev/build_variant.sh anchor "-DTC_ANCHOR_STATS" icp; run with
TC_HIP_LIB=threecrate_amd/variants/libthreecrate_hip_anchor.so).  The library prints, per ICP call, which share of the source points
the inscribed-ball test keeps, which share a 2-NN anchor (runner-up bound of the last search minus the motion since) would certify,
and -- what decides the time -- the share of WAVE TRIPS in which all 64 lanes are certified (only those skip the search)."""
import numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
n = 1000000
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(n, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
dt, ds = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
th, sh = tc.Cloud(ctx, dt), tc.Cloud(ctx, ds)
th.estimate_normals(16, out=False)
for iters, label in ((6, "iterations 1-6"), (12, "iterations 1-12"), (24, "iterations 1-24"), (36, "iterations 1-36"), (50, "iterations 1-50")):
    print("==", label, flush=True)
    r = sh.icp_point_to_plane(th, None, iters, None, 0.0)
print("== 8 iterations from the converged transform", flush=True)
sh.icp_point_to_plane(th, r.transformation, 8, None, 0.0)
