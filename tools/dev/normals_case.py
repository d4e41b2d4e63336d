import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: one case of normals_fuzz.py in detail: python tools/dev/normals_case.py <seed> <case> <point>"""
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from tests import h1
seed, case, pt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng([seed, case])
n = int(rng.choice([2, 3, 7, 40, 300, 1500, 6000])); kind = int(rng.integers(0, 6))
if kind == 0: p = rng.random((n, 3))
elif kind == 1: p = rng.random((n, 3)) * np.array([10.0, 3.0, 0.2])
elif kind == 2: u = rng.random((n, 2)); p = np.stack([u[:, 0], u[:, 1], 0.1 * np.sin(6 * u[:, 0]) + 1e-4 * rng.normal(size=n)], 1)
elif kind == 3: t = rng.random(n); p = np.stack([t, 2 * t, -t], 1) + 1e-3 * rng.normal(size=(n, 3))
elif kind == 4: p = np.round(rng.random((n, 3)) * 8) / 8
else: p = rng.random((n, 3)); p[: n // 4] = p[0]
p = (p * rng.choice([1e-2, 1.0, 50.0])).astype(np.float32)
spacing = float(np.linalg.norm(p.max(0) - p.min(0))) / max(n, 2) ** (1.0 / 3.0) + 1e-12
k = int(rng.choice([1, 2, 3, 5, 10, 16, 17, 31, 32, 33, 64, 100, 127, 128, int(rng.integers(1, 129))]))
radius = None if rng.random() < 0.6 else float(spacing * rng.choice([0.05, 0.5, 1.5, 4.0, 50.0]))
orient = bool(rng.random() < 0.7)
vp = None if rng.random() < 0.6 else tuple((p.mean(0) + (p.std(0) + 1e-3) * rng.normal(0, 3, 3)).tolist())
print("n", n, "kind", kind, "k", k, "radius", radius, "orient", orient, "vp", vp)
ctx = tc.GpuContext(0)
cfg = tc.NormalEstimationConfig(k_neighbors=k, radius=radius, consistent_orientation=orient, viewpoint=vp)
g = ctx.estimate_normals_with_config(p, cfg)
r = O.estimate_normals(p, k, radius=radius, consistent_orientation=orient, viewpoint=vp)
print("gpu   ", g[pt]); print("oracle", r[pt])
idx, dist, cnt = O.knn_batch(p, p[pt:pt + 1], min(k + 6, n))
print("oracle k-NN idx ", idx[0, :cnt[0]]); print("          dist2", dist[0, :cnt[0]])
gi, gd, gc = ctx.find_k_nearest_batch(p, p[pt:pt + 1], min(k + 6, n))
print("gpu    k-NN idx ", gi[0, :gc[0]]); print("          dist ", gd[0, :gc[0]])
nb = [int(i) for i in idx[0, :k + 1]]
if pt not in nb: print("self not among the k+1 nearest returned by the oracle's tree!")
P = p[nb].astype(np.float64)
C = np.cov(P.T, bias=True)
w, v = np.linalg.eigh(C)
print("cov\n", C, "\neig", w, "\nvec0", v[:, 0])
print("duplicates of the query:", int((p == p[pt]).all(1).sum()))
print("reference solver spread over rounding-level variants of this neighbourhood's covariance:", h1.reference_solver_spread(p[nb]))
a, b = set(int(i) for i in idx[0, :k + 1]), set(int(i) for i in gi[0, :k + 1])
print("k+1 nearest: oracle-only", sorted(a - b), "gpu-only", sorted(b - a))
print("sorted d2 (f32 formula) of all points, entries k-2..k+3:", np.sort(h1.d2_f32(p, p[pt]))[max(k - 2, 0):k + 4])
# what each side's normal corresponds to: the reference formula on the oracle's / the GPU's neighbour set
for name, ids in (("oracle set", idx[0, :k + 1]), ("gpu set", gi[0, :k + 1])):
    Q = p[[int(i) for i in ids]].astype(np.float64)
    w, v = np.linalg.eigh(np.cov(Q.T, bias=True))
    print(name, "f64 normal", v[:, 0], "| cos to gpu", abs(float(v[:, 0] @ g[pt, 3:])), "cos to oracle", abs(float(v[:, 0] @ r[pt, 3:])))
