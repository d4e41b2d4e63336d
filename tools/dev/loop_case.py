import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: replay one case of loop_fuzz.py iteration by iteration: python tools/dev/loop_case.py <seed> <case>"""
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth
import importlib.util
spec = importlib.util.spec_from_file_location("lf", os.path.join(os.path.dirname(os.path.abspath(__file__)), "loop_fuzz.py"))
lf = importlib.util.module_from_spec(spec); spec.loader.exec_module(lf)
seed, case = int(sys.argv[1]), int(sys.argv[2])
ctx = tc.GpuContext(0)
lf.run(1e9, seed, ctx, only_case=case)
# regenerate the inputs exactly as run() does
rng = np.random.default_rng([seed, case])
n = int(rng.choice([300, 1500, 4000])); kind = int(rng.integers(0, 3))
if kind == 0: tgt = rng.random((n, 3))
elif kind == 1: u = rng.random((n, 2)); tgt = np.stack([u[:, 0], u[:, 1], 0.15 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])], 1)
else: tgt = rng.random((n, 3)) * np.array([4.0, 1.0, 0.3])
tgt = (tgt * rng.choice([0.1, 1.0, 20.0])).astype(np.float32)
ext = float(np.linalg.norm(tgt.max(0) - tgt.min(0)))
T = synth.yaw_isometry(tuple((rng.normal(0, 0.03, 3) * ext).tolist()), float(rng.normal(0, 0.05)))
m = int(n * rng.choice([0.3, 1.0]))
src = synth.apply_isometry(T, tgt[rng.permutation(n)[:m]])
if rng.random() < 0.5: src = (src + rng.normal(0, 2e-3 * ext, src.shape)).astype(np.float32)
p2plane = bool(rng.random() < 0.5); iters = int(rng.integers(1, 61))
thr = float(rng.choice([0.0, 1e-12, 1e-9, 1e-6, 1e-4, 1e-2])) * (ext * ext if rng.random() < 0.5 else 1.0)
md = None if rng.random() < 0.5 else float(ext * rng.choice([0.05, 0.2, 1.0]))
init = None if rng.random() < 0.5 else synth.yaw_isometry(tuple((rng.normal(0, 0.01, 3) * ext).tolist()), float(rng.normal(0, 0.01)))
print("n", n, "m", m, "ext", ext, "p2plane", p2plane, "iters", iters, "thr", thr, "md", md)
nrm = O.estimate_normals(tgt, min(10, n - 1))[:, 3:] if p2plane else None
for k in range(1, iters + 1):
    if p2plane:
        g = ctx.icp_point_to_plane_detailed(src, tgt, nrm, init, k, md, thr); r = O.icp_point_to_plane_detailed(src, tgt, nrm, init, k, md, thr)
    else:
        g = ctx.icp_detailed(src, tgt, init, k, md, thr); r = O.icp_detailed(src, tgt, init, k, md, thr)
    nd = int((g.correspondences[:, 1] != r.correspondences[:, 1]).sum()) if len(g.correspondences) == len(r.correspondences) else -1
    print(f"it {k:2d} gpu conv {g.converged} {g.iterations} mse {g.mse:.9e} | oracle conv {r.converged} {r.iterations} mse {r.mse:.9e} | frob {lf.frob(g.transformation, r.transformation):.3e} corr diff {nd} of {len(r.correspondences)}")
    if g.converged and r.converged: break
