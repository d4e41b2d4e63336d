import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: replay ONE case of tools/dev/paths_stress.py (same generator, same seed: the earlier cases are generated, not run) and print
what the plain-call road and the handle road returned:  python tools/dev/paths_case.py <seed> <case> [<case> ...]"""
import importlib.util, numpy as np, torch
import threecrate_amd as tc
from threecrate_amd import synth
src_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "paths_stress.py")
code = open(src_path).read().replace("\nmain()\n", "\n")
ns = {"__name__": "paths_stress_lib", "__file__": src_path}
exec(compile(code, src_path, "exec"), ns)
seed, wanted = int(sys.argv[1]), sorted(int(a) for a in sys.argv[2:])
ctx = tc.GpuContext(0)
rng = np.random.default_rng(seed)
case = 0
while wanted:
    case += 1
    kind, p = ns["cloud"](rng)
    ext = float(np.linalg.norm(np.percentile(p, 99, 0) - np.percentile(p, 1, 0)))
    T = synth.yaw_isometry(tuple((rng.normal(0, 0.004, 3) * ext).tolist()), float(rng.normal(0, 0.01)))
    src = synth.apply_isometry(T, p[rng.permutation(len(p))[: max(3, int(len(p) * rng.choice([0.3, 1.0])))]])
    k = int(rng.choice([5, 10, 16, 24])); iters = int(rng.integers(1, 14))
    md = None if rng.random() < 0.6 else float(ext * rng.choice([0.02, 0.2]))
    p2plane = bool(rng.random() < 0.7)
    if case != wanted[0]:
        continue
    wanted.pop(0)
    out = ns["run_paths"](ctx, p, src, k, iters, md, p2plane)
    a, b = out["device"], out["handles"]
    dT = float(np.abs(np.asarray(a[1], np.float64) - np.asarray(b[1], np.float64)).max())
    rows = int((a[4] != b[4]).reshape(len(a[4]), -1).any(1).sum()) if a[4].shape == b[4].shape else -1
    print(f"case {case}: kind {kind} n {len(p)} m {len(src)} k {k} iters {iters} md {md} p2plane {p2plane} ext {ext:.3f}")
    print(f"   plain: mse {float(a[2]):.9e} iterations {a[3]} pairs {len(a[4])} | handles: mse {float(b[2]):.9e} iterations {b[3]} pairs {len(b[4])}")
    print(f"   |dT|max {dT:.3e}  rel mse diff {abs(float(a[2]) - float(b[2])) / max(float(a[2]), 1e-30):.3e}  pair rows differing {rows}  normals rows differing {int((a[0] != b[0]).any(1).sum())}")
    if rows > 0:            # the rows the two roads resolved differently: the source point's distance to either target under the plain road's transform of the last search
        from oracle import oracle as O_
        A, B = np.asarray(a[4]), np.asarray(b[4])
        idx = np.nonzero((A != B).reshape(len(A), -1).any(1))[0]
        # (under the transform the last iteration SEARCHED with: the same call stopped one iteration earlier)
        Tprev = O_.IDENTITY
        if iters > 1:
            d_, ds_ = torch.from_numpy(p).cuda(), torch.from_numpy(src).cuda()
            Tprev = (ctx.icp_point_to_plane_detailed(ds_, d_, ctx.estimate_normals(d_, k), None, iters - 1, md, 0.0, correspondences=False) if p2plane
                     else ctx.icp_detailed(ds_, d_, None, iters - 1, md, 0.0, correspondences=False)).transformation
        q = O_.isometry_apply(Tprev, src[A[idx, 0]]).astype(np.float64)
        da = np.linalg.norm(q - p[A[idx, 1]].astype(np.float64), axis=1); db = np.linalg.norm(q - p[B[idx, 1]].astype(np.float64), axis=1)
        print("   differing rows: source, target (plain), target (handles), distance to either, difference:")
        for r_, i_ in enumerate(idx[:24]):
            print(f"      {int(A[i_, 0]):8d} {int(A[i_, 1]):8d} {int(B[i_, 1]):8d}  {da[r_]:.9e} {db[r_]:.9e}  {abs(da[r_] - db[r_]):.2e}   same source row: {bool(A[i_, 0] == B[i_, 0])}")
    for it in range(1, iters + 1):          # where the two roads part: the same registration stopped after 1 .. iters iterations
        d, ds = torch.from_numpy(p).cuda(), torch.from_numpy(src).cuda()
        nrm = ctx.estimate_normals(d, k)
        r1 = ctx.icp_point_to_plane_detailed(ds, d, nrm, None, it, md, 0.0) if p2plane else ctx.icp_detailed(ds, d, None, it, md, 0.0)
        t, s = tc.Cloud(ctx, d), tc.Cloud(ctx, ds); t.estimate_normals(k, out=False)
        r2 = s.icp_point_to_plane(t, None, it, md, 0.0) if p2plane else s.icp_detailed(t, None, it, md, 0.0)
        t.close(); s.close()
        print(f"   after {it:2d} iterations: mse {r1.mse:.9e} / {r2.mse:.9e}   |dT|max {float(np.abs(np.asarray(r1.transformation, np.float64) - np.asarray(r2.transformation, np.float64)).max()):.2e}")
