import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: randomized differential run GPU vs oracle on small, odd clouds (exactness of the searches, no crashes).
usage: python tools/dev/fuzz.py [seconds] [seed]      (FUZZ_BIG=1: clouds of 40 k .. 400 k points)
tests/test_gpu_fuzz.py calls run() in-process."""
import time
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth

import importlib.util as _ilu
_spec = _ilu.spec_from_file_location("tc_normals_fuzz_", os.path.join(os.path.dirname(os.path.abspath(__file__)), "normals_fuzz.py"))
_nf = _ilu.module_from_spec(_spec); _spec.loader.exec_module(_nf)
explain_offender = _nf.explain_offender

SMALL = [5, 17, 64, 300, 2000, 9000]
BIG = [40000, 120000, 270000, 400000]


def run(budget, seed, ctx, sizes=SMALL, log=print, min_cases=0):
    rng = np.random.default_rng(seed)
    def cloud(kind, n):
        if kind == 0: p = rng.random((n, 3))
        elif kind == 1: p = rng.random((n, 3)) * np.array([10.0, 3.0, 0.2])                   # slab
        elif kind == 2: p = np.concatenate([rng.normal(0, 0.05, (n // 2, 3)), rng.normal(3, 0.2, (n - n // 2, 3))])   # clusters
        elif kind == 3: u = rng.random((n, 2)); p = np.stack([u[:, 0], u[:, 1], 0.1 * np.sin(6 * u[:, 0]) + 1e-4 * rng.normal(size=n)], 1)   # surface
        elif kind == 4: t = rng.random(n); p = np.stack([t, 2 * t, -t], 1) + 1e-3 * rng.normal(size=(n, 3))   # near-collinear
        elif kind == 5: p = np.round(rng.random((n, 3)) * 8) / 8 + 1e-5 * rng.normal(size=(n, 3))   # lattice-ish with near duplicates
        else: p = rng.random((n, 3)); p[: n // 4] = p[0]                                        # many exact duplicates
        p = p * rng.choice([1e-2, 1.0, 50.0])
        if n >= 64 and rng.random() < 0.35:            # a few far outliers: the grid's box gets clamped (GridGeom::clamped)
            m = int(rng.integers(1, 6))
            ext = p.max(0) - p.min(0) + 1e-9
            far = p.mean(0) + ext * rng.choice([-1.0, 1.0], (m, 3)) * rng.uniform(3, 200, (m, 3)) * (rng.random((m, 3)) < 0.6)
            if m >= 2: far[1] = far[0] + ext * 1e-3     # two outliers next to each other: neighbours of one another
            p[rng.integers(0, n, m)] = far
        return p.astype(np.float32)

    t_end = time.time() + budget
    t_hard = t_end + 7 * budget          # (min_cases: a slow or cold box goes on past the budget until it has that many cases)
    cases = bad = 0
    while time.time() < t_end or (cases < min_cases and time.time() < t_hard):
        kind = int(rng.integers(0, 7)); n = int(rng.choice(sizes)); cases += 1
        tgt = cloud(kind, n)
        tag = f"case {cases} kind {kind} n {n}"
        try:
            # k-NN distances (exact) on inside / outside queries
            k = int(rng.choice([1, 3, 8, 16, 40]))
            qs = np.concatenate([tgt[rng.integers(0, n, 20)], (tgt.mean(0) + tgt.std(0) * 4 * rng.normal(size=(10, 3))).astype(np.float32)])
            gi, gd, gc = ctx.find_k_nearest_batch(tgt, qs, k)
            oi, od, oc = O.knn_batch(tgt, qs, k)
            if not (np.array_equal(gc, oc) and all(np.array_equal(gd[q, :gc[q]], od[q, :oc[q]]) for q in range(len(qs)))):
                bad += 1; log("KNN MISMATCH", tag, k)
            # one ICP iteration from a random small motion: the correspondences are the exact 1-NN
            ext = float(np.linalg.norm(tgt.max(0) - tgt.min(0))) + 1e-6
            T = synth.yaw_isometry(tuple((rng.normal(0, 0.02, 3) * ext).tolist()), float(rng.normal(0, 0.03)))
            src = synth.apply_isometry(T, tgt[rng.permutation(n)[: max(3, n // 2)]])
            md = None if rng.random() < 0.5 else float(ext * rng.choice([0.01, 0.1, 1.0]))
            try:
                g = ctx.icp_detailed(src, tgt, None, 1, md, 0.0)
                gerr = None
            except tc.Error as e:
                g, gerr = None, type(e).__name__
            try:
                r = O.icp_detailed(src, tgt, None, 1, md, 0.0)
                rerr = None
            except O.OracleError as e:
                r, rerr = None, "err"
            if (g is None) != (r is None):
                bad += 1; log("ICP ERROR MISMATCH", tag, gerr, rerr, md)
            elif g is not None:
                same = len(g.correspondences) == len(r.correspondences) and np.array_equal(g.correspondences[:, 0], r.correspondences[:, 0])
                if same:
                    # targets may differ only where the distances tie
                    diff = np.nonzero(g.correspondences[:, 1] != r.correspondences[:, 1])[0]
                    for i in diff[:50]:
                        s = src[g.correspondences[i, 0]].astype(np.float32)
                        da = np.sum((tgt[g.correspondences[i, 1]] - s) ** 2, dtype=np.float32); db = np.sum((tgt[r.correspondences[i, 1]] - s) ** 2, dtype=np.float32)
                        if abs(float(da) - float(db)) > 1e-6 * max(float(db), 1e-30): same = False
                if not same:
                    bad += 1; log("ICP CORRESPONDENCE MISMATCH", tag, md, len(g.correspondences), len(r.correspondences))
            # voxel filter (bit exact)
            vs = float(ext * rng.choice([0.02, 0.1, 0.5]))
            try:
                gv = ctx.voxel_grid_filter(tgt, vs); ov = O.voxel_grid_filter(tgt, vs)
                if not np.array_equal(gv, ov): bad += 1; log("VOXEL MISMATCH", tag, vs)
            except tc.Unsupported:
                pass
            # normals: valid unit vectors, parity where the neighbourhood is well conditioned is covered by the tests
            if n >= 17:
                gn = ctx.estimate_normals(tgt, min(16, n - 1))
                nn = np.linalg.norm(gn[:, 3:], axis=1)
                if not (np.isfinite(gn).all() and np.abs(nn - 1).max() < 1e-4): bad += 1; log("NORMALS INVALID", tag)
                # same eigen algorithm as the reference: where the neighbour SETS are untied the normals agree to the last bits,
                # ill-conditioned neighbourhoods included (kinds with exact duplicates / lattices have tied neighbours)
                if kind <= 4 and n <= 9000:
                    rn = O.estimate_normals(tgt, min(16, n - 1))
                    c = np.abs((gn[:, 3:].astype(np.float64) * rn[:, 3:].astype(np.float64)).sum(1))
                    off = np.nonzero(~(c >= 1 - 1e-4))[0]
                    if len(off):
                        # every offender must be explained by the input (tools/dev/normals_fuzz.py: boundary tie, degenerate
                        # eigen-pair, discontinuity of the reference's eigen-solver)
                        kk = min(16, n - 1)
                        tree = O.KdTree(tgt)
                        why = [explain_offender(tgt, int(i_), kk, None, tree) for i_ in off[:50]]
                        if not all(ok for ok, _ in why): bad += 1; log("NORMALS PARITY", tag, len(off), "of", n, [w for ok, w in why if not ok][:2])
        except Exception as e:
            bad += 1; log("EXCEPTION", tag, type(e).__name__, e)
    log(f"fuzz: {cases} cases, {bad} problems")
    return cases, bad


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    run(budget, seed, tc.GpuContext(0), BIG if os.environ.get("FUZZ_BIG") else SMALL)
