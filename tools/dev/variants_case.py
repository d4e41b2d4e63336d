import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: one KISS-ICP case of variants_fuzz.py iteration by iteration: python tools/dev/variants_case.py <seed> <case>"""
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth
import importlib.util
spec = importlib.util.spec_from_file_location("vf", os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants_fuzz.py"))
vf = importlib.util.module_from_spec(spec); spec.loader.exec_module(vf)
seed, case = int(sys.argv[1]), int(sys.argv[2])
ctx = tc.GpuContext(0)
vf.run(1e9, seed, ctx, only_case=case)
rng = np.random.default_rng([seed, case])
n = int(rng.choice([600, 2500, 8000])); kind = int(rng.integers(0, 3))
if kind == 0: tgt = rng.random((n, 3))
elif kind == 1: u = rng.random((n, 2)); tgt = np.stack([u[:, 0], u[:, 1], 0.15 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])], 1)
else: tgt = rng.random((n, 3)) * np.array([4.0, 1.0, 0.3])
scale = float(rng.choice([1.0, 10.0, 40.0])); tgt = (tgt * scale).astype(np.float32)
ext = float(np.linalg.norm(tgt.max(0) - tgt.min(0))); spacing = ext / n ** (1.0 / 3.0)
T = synth.yaw_isometry(tuple((rng.normal(0, 0.01, 3) * ext).tolist()), float(rng.normal(0, 0.02)))
src = synth.apply_isometry(T, tgt[rng.permutation(n)[: int(n * rng.choice([0.5, 1.0]))]])
if rng.random() < 0.5: src = (src + rng.normal(0, 1e-3 * ext, src.shape)).astype(np.float32)
init = None if rng.random() < 0.6 else synth.yaw_isometry(tuple((rng.normal(0, 0.004, 3) * ext).tolist()), float(rng.normal(0, 0.005)))
which = int(rng.integers(0, 3))
assert which == 1, "KISS-ICP cases only"
f_ = 0.3 / spacing
tgt = (tgt * f_).astype(np.float32); src = (src * f_).astype(np.float32)
if init is not None: init = np.concatenate([init[:4], init[4:] * f_]).astype(np.float32)
ext *= f_; spacing *= f_
vs = float(spacing * rng.choice([0.8, 1.5, 3.0])); mx = float(ext * rng.choice([0.5, 2.0])); mn = float(ext * rng.choice([0.0, 0.05]))
it = int(rng.integers(1, 61))
print("n", n, "m", len(src), "ext", ext, "voxel", vs, "range", mn, mx, "iters", it)
for k in range(1, it + 1):
    g = ctx.kiss_icp(src, tgt, init, tc.KissIcpConfig(voxel_size=vs, max_range=mx, min_range=mn, max_iterations=k))
    r, nd = O.kiss_icp(src, tgt, init, vs, mx, mn, k)
    same = len(g.correspondences) == len(r.correspondences)
    ndiff = int((g.correspondences != r.correspondences).any(axis=1).sum()) if same else -1
    print(f"it {k:2d} gpu {g.converged} {g.iterations} mse {g.mse:.9e} | oracle {r.converged} {r.iterations} mse {r.mse:.9e} | frob {vf.frob(g.transformation, r.transformation):.3e} pairs {len(g.correspondences)}/{len(r.correspondences)} differing {ndiff}")
    if g.converged and r.converged: break
