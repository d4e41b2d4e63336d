import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: replay ONE case of normals_fuzz.py and print, for every unexplained offender, both sides' neighbourhoods:
usage: python tools/dev/ncase.py <seed> <case>"""
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from tests import h1
from tools.dev.normals_fuzz import explain_offender

seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, case])
n = int(rng.choice([2, 3, 7, 40, 300, 1500, 6000]))
kind = int(rng.integers(0, 6))
if kind == 0: p = rng.random((n, 3))
elif kind == 1: p = rng.random((n, 3)) * np.array([10.0, 3.0, 0.2])
elif kind == 2: u = rng.random((n, 2)); p = np.stack([u[:, 0], u[:, 1], 0.1 * np.sin(6 * u[:, 0]) + 1e-4 * rng.normal(size=n)], 1)
elif kind == 3: t = rng.random(n); p = np.stack([t, 2 * t, -t], 1) + 1e-3 * rng.normal(size=(n, 3))
elif kind == 4: p = np.round(rng.random((n, 3)) * 8) / 8
else: p = rng.random((n, 3)); p[: n // 4] = p[0]
p = (p * rng.choice([1e-2, 1.0, 50.0])).astype(np.float32)
spacing = float(np.linalg.norm(p.max(0) - p.min(0))) / max(n, 2) ** (1.0 / 3.0) + 1e-12
k = int(rng.choice([1, 2, 3, 5, 10, 16, 17, 31, 32, 33, 64, 100, 127, 128, int(rng.integers(1, 129)), 129, 200, int(rng.integers(129, 600))]))
radius = None if rng.random() < 0.6 else float(spacing * rng.choice([0.05, 0.5, 1.5, 4.0, 50.0]))
if k > 128: radius = None
orient = bool(rng.random() < 0.7)
vp = None if rng.random() < 0.6 else tuple((p.mean(0) + (p.std(0) + 1e-3) * rng.normal(0, 3, 3)).tolist())
mode = int(rng.integers(0, 3))
print(f"n {n} kind {kind} k {k} radius {radius} orient {orient} vp {vp} mode {mode} scale-max {p.max():.4g}")
np.save("/tmp/ncase_points.npy", p)
ctx = tc.GpuContext(0)
cfg = tc.NormalEstimationConfig(k_neighbors=k, radius=radius, consistent_orientation=orient, viewpoint=vp)
g = ctx.estimate_normals_with_config(p, cfg)
import torch
g1 = ctx.estimate_normals_with_config(torch.from_numpy(p).cuda(), cfg).cpu().numpy()
hh = tc.Cloud(ctx, p); g2 = hh.estimate_normals(k, cfg); hh.close()
print("plain vs device tensor identical:", np.array_equal(g, g1), " plain vs handle identical:", np.array_equal(g, g2))
r = O.estimate_normals(p, k, radius=radius, consistent_orientation=orient, viewpoint=vp)
c = h1.cos_abs(g[:, 3:], r[:, 3:])
off = np.nonzero(~(c >= 1 - 1e-4))[0]
tree = O.KdTree(p)
print(len(off), "offenders")
shown = explained_n = 0
for i in off:
    ok, why = explain_offender(p, int(i), k, radius, tree, float(1 - c[i]), g[i, 3:])
    if ok:
        explained_n += 1
        if explained_n <= 2: print(f"explained: point {i}: {why}")
        continue
    shown += 1
    if shown > 3: break
    print(f"--- point {i} {p[i]} 1-|cos| {1 - c[i]:.3g}: {why}")
    print("gpu normal", g[i, 3:], "oracle normal", r[i, 3:])
    oi, od = tree.find_k_nearest(p[i], k + 1)
    gl = ctx.find_k_nearest(p, p[i], k + 1)
    gi, gd = [a for a, _ in gl], [b for _, b in gl]
    d2 = h1.d2_f32(p, p[i])
    print("oracle k+1:", [(int(a), float(np.float32(b))) for a, b in zip(oi, od)])
    print("gpu    k+1:", [(int(a), float(np.float32(b))) for a, b in zip(gi, gd)])
    order = np.argsort(d2, kind="stable")[: k + 6]
    print("brute d2 (f32) first k+6:", [(int(a), float(d2[a])) for a in order])
    print("set difference oracle-gpu:", sorted(set(map(int, oi)) - set(map(int, gi))), "gpu-oracle:", sorted(set(map(int, gi)) - set(map(int, oi))))
    # the normal of each side's set by a float64 PCA (which set explains which normal)
    for name, idx in (("oracle set", oi), ("gpu set", gi)):
        nb = [int(a) for a in idx if int(a) != i][:k] + [int(i)]
        ev, evec = np.linalg.eigh(np.cov(p[nb].astype(np.float64).T, bias=True))
        print(name, "f64 PCA normal", evec[:, 0], "eigenvalues", ev)
