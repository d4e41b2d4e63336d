import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: ONE input through every way of calling the library, in one context, on a stream of clouds of changing size -- the answers
must not depend on the road taken.  Bit for bit: host arrays vs device tensors, polled pinned words vs copies + stream
synchronisations (TC_NO_PINNED_POLL), profiling off / every kernel / sampled, binned vs atomic index placement, a second call on the
same buffers.  Handles share ONE grid between normals and registration (another cell edge): normals and correspondences equal
up to exact distance ties between points of different cells (broken by position: a handful of rows per million), the transform
to 1e-6 (another source order = another summation order).
usage: python tools/dev/paths_stress.py [seconds] [seed]"""
import time
import numpy as np, torch
import threecrate_amd as tc
from threecrate_amd import synth


def cloud(rng):
    n = int(rng.choice([900, 5_000, 40_000, 130_000, 262_143, 262_144, 300_000, 524_288, 1_000_000]))
    kind = int(rng.integers(0, 4))
    if kind == 0: p = synth.uniform_cloud(n, seed=int(rng.integers(1, 10_000)))
    elif kind == 1:
        u = rng.random((n, 2)); p = np.stack([u[:, 0], u[:, 1], 0.1 * np.sin(5 * u[:, 0]) * np.cos(3 * u[:, 1]) + 2e-4 * rng.standard_normal(n)], 1).astype(np.float32)
    elif kind == 2: p = (rng.random((n, 3)) * np.array([6.0, 2.0, 0.4])).astype(np.float32)
    else:
        p = synth.uniform_cloud(n, seed=int(rng.integers(1, 10_000)))
        p[rng.integers(0, n, max(1, n // 5000))] *= np.float32(30.0)          # far outliers: the robust box
    return kind, np.ascontiguousarray(p, dtype=np.float32)


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def run_paths(ctx, p, src, k, iters, md, p2plane):
    """-> {road: (normals, T, mse, iterations, pairs)}"""
    out = {}
    d, ds = torch.from_numpy(p).cuda(), torch.from_numpy(src).cuda()
    def plain(device):
        P, S = (d, ds) if device else (p, src)
        nrm = ctx.estimate_normals(P, k)
        r = (ctx.icp_point_to_plane_detailed(S, P, nrm, None, iters, md, 0.0, correspondences=True) if p2plane
             else ctx.icp_detailed(S, P, None, iters, md, 0.0, correspondences=True))
        return (nrm.cpu().numpy() if device else nrm, r.transformation, np.float32(r.mse), r.iterations, np.asarray(r.correspondences))
    out["device"] = plain(True)
    out["host"] = plain(False)
    out["device again"] = plain(True)
    for name, env in (("no pinned poll", {"TC_NO_PINNED_POLL": "1"}), ("atomic index", {"TC_INDEX_BINNED": "0"})):
        os.environ.update(env)
        try: out[name] = plain(True)
        finally:
            for e in env: os.environ.pop(e)
    for mode in (1, 2):
        ctx.profile_enable(mode)
        try: out[f"profile {mode}"] = plain(True)
        finally: ctx.profile_enable(0); ctx.profile_reset()
    t, s = tc.Cloud(ctx, d), tc.Cloud(ctx, ds)
    try:
        nrm = t.estimate_normals(k)
        r = (s.icp_point_to_plane(t, None, iters, md, 0.0, correspondences=True) if p2plane else s.icp_detailed(t, None, iters, md, 0.0, correspondences=True))
        out["handles"] = (nrm.cpu().numpy(), r.transformation, np.float32(r.mse), r.iterations, np.asarray(r.correspondences))
    finally:
        t.close(); s.close()
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = tc.GpuContext(0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = bad = ties = 0
    while time.time() < t_end:
        cases += 1
        kind, p = cloud(rng)
        ext = float(np.linalg.norm(np.percentile(p, 99, 0) - np.percentile(p, 1, 0)))
        T = synth.yaw_isometry(tuple((rng.normal(0, 0.004, 3) * ext).tolist()), float(rng.normal(0, 0.01)))
        src = synth.apply_isometry(T, p[rng.permutation(len(p))[: max(3, int(len(p) * rng.choice([0.3, 1.0])))]])
        k = int(rng.choice([5, 10, 16, 24]))
        iters = int(rng.integers(1, 14))
        md = None if rng.random() < 0.6 else float(ext * rng.choice([0.02, 0.2]))
        p2plane = bool(rng.random() < 0.7)
        tag = f"case {cases}: kind {kind} n {len(p)} m {len(src)} k {k} iters {iters} md {md} {'p2plane' if p2plane else 'p2p'}"
        try:
            out = run_paths(ctx, p, src, k, iters, md, p2plane)
        except tc.Error as e:
            print("ERROR", tag, type(e).__name__, str(e)[:120], flush=True); bad += 1; continue
        ref = out["device"]
        for road, got in out.items():
            if road == "device": continue
            names = ("normals", "T", "mse", "iterations", "pairs")
            for what, a, b in zip(names, ref, got):
                if road == "handles" and what in ("T", "mse", "normals", "pairs"):
                    # another grid: an exact distance tie between points of different cells is broken by position, i.e. by the grid
                    # (the reference's own tie order is its heap's); a handful of rows per million may pick the other neighbour
                    # (T: 1e-6 when the two roads chose the same pairs throughout the last iteration; where tie rows differ the roads are
                    # on two trajectories, a few 1e-6 apart while the registration is still descending -- campaign 504 case 1115: 46 of
                    # 300 000 rows, |dT| 2.1e-6 after 10 of 10 iterations, tools/dev/paths_case.py prints it iteration by iteration)
                    if what == "T":
                        tie_rows = int((np.asarray(ref[4]) != np.asarray(got[4])).reshape(len(ref[4]), -1).any(1).sum()) if len(ref[4]) == len(got[4]) else 0
                        ok = np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=0, atol=(5e-6 if tie_rows else 1e-6) * max(1.0, ext))
                    # (a converged noise-free pair: mse ~ rounding.  A run stopped WHILE it converges -- the mse falling 10-50x per
                    # iteration -- returns the mse measured under the previous iteration's transform, and the two roads' transforms
                    # differ by up to ~2e-5 there (another grid = another tie order = another trajectory; they meet again at 1e-7):
                    # d mse ~ 2 sqrt(mse) |dT| extent.  tools/dev/paths_case.py replays a case iteration by iteration; campaign 501.)
                    elif what == "mse": ok = abs(float(a) - float(b)) <= 1e-4 * max(float(a), 1e-30) + 1e-10 * ext * ext + 2.0 * float(a) ** 0.5 * 3e-5 * ext
                    else:
                        rows = (np.asarray(a) != np.asarray(b)).reshape(len(a), -1).any(1).sum()
                        ties += int(rows)
                        ok = rows <= max(8, len(a) // 5_000)          # (+ near-tie pairs that flip with the 1e-7 the transforms differ by)
                        if not ok and what == "pairs" and rows <= max(8, len(a) // 500):
                            # more rows than that (campaign 506 case 296: 19 of 78 643, twelve p2p iterations into a descent): accepted
                            # when every one of them IS a near-tie -- the source point's distances to the two targets, under the
                            # transform the LAST iteration searched with (the same call stopped one iteration earlier), differ by less
                            # than the two roads' transforms move it apart
                            from oracle import oracle as O_
                            import torch as torch_
                            A, B = np.asarray(a), np.asarray(b)
                            idx = np.nonzero((A != B).reshape(len(A), -1).any(1))[0]
                            Tprev = O_.IDENTITY
                            if iters > 1:
                                dd, dss = torch_.from_numpy(p).cuda(), torch_.from_numpy(src).cuda()
                                rp = (ctx.icp_point_to_plane_detailed(dss, dd, ctx.estimate_normals(dd, k), None, iters - 1, md, 0.0, correspondences=False) if p2plane
                                      else ctx.icp_detailed(dss, dd, None, iters - 1, md, 0.0, correspondences=False))
                                Tprev = rp.transformation
                            q = O_.isometry_apply(Tprev, src[A[idx, 0]]).astype(np.float64)
                            da = np.linalg.norm(q - p[A[idx, 1]].astype(np.float64), axis=1)
                            db = np.linalg.norm(q - p[B[idx, 1]].astype(np.float64), axis=1)
                            dT = float(np.abs(np.asarray(ref[1], np.float64) - np.asarray(got[1], np.float64)).max())
                            ok = bool((A[idx, 0] == B[idx, 0]).all() and (np.abs(da - db) <= 4.0 * (dT + 1e-7) * max(1.0, ext)).all())
                else:
                    ok = same(a, b)
                if not ok:
                    bad += 1
                    nd = int((np.asarray(a) != np.asarray(b)).sum()) if np.asarray(a).shape == np.asarray(b).shape else -1
                    print(f"DIFFERENT {tag} | road '{road}': {what} differs ({nd} elements)", flush=True)
                    break
    print(f"paths stress: {cases} cases, {bad} differences; handles vs plain calls: {ties} rows resolved the other way (exact ties across cells)")


main()
