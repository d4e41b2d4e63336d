import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faulthandler; faulthandler.dump_traceback_later(120, exit=True)
import time, numpy as np, torch, threecrate_amd as tc
from threecrate_amd import synth
from oracle import oracle as O
ctx = tc.GpuContext(0)
src, tgt, T = synth.registration_pair(20000, seed=1)
t0 = time.time(); g = ctx.gicp(src, tgt, None, tc.GicpConfig(30, 1.0, 1e-6, 20)); t1 = time.time()
r = O.gicp(src, tgt, None, 30, 1.0, 1e-6, 20)
print("gpu", t1 - t0, g.iterations, g.converged, g.mse, g.transformation)
print("ref", r.iterations, r.converged, r.mse, r.transformation)
print("frob", np.linalg.norm(O.isometry_to_matrix(g.transformation).astype(np.float64) - O.isometry_to_matrix(r.transformation)))
print("corr mismatch", (g.correspondences != r.correspondences).any(axis=1).mean() if len(g.correspondences) == len(r.correspondences) else (len(g.correspondences), len(r.correspondences)))
