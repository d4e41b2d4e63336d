import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: randomized differential run of the ICP LOOP CONTROL against the oracle: iteration counts 1..60, convergence thresholds
from 0 to 1e-2 (runs that stop after any number of iterations, in the middle of any chunk of the enqueue schedule), with and
without a maximum correspondence distance, random initial guesses, point-to-point and point-to-plane, source and target of
different sizes, plain calls and cloud handles.  Compared: converged flag, iteration count, transform (1e-5 Frobenius, or the
oracle's own order noise where the pair amplifies rounding), mse, number of correspondences.
usage: python tools/dev/loop_fuzz.py [seconds] [seed];  tests/test_gpu_fuzz.py calls run() in-process."""
import time
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth


def frob(a, b):
    return float(np.linalg.norm(O.isometry_to_matrix(a).astype(np.float64) - O.isometry_to_matrix(b).astype(np.float64)))


EPS32 = float(np.finfo(np.float32).eps)


def d2_f32(a, b):
    d = (np.asarray(a, np.float32) - np.asarray(b, np.float32)).astype(np.float32)
    x, y, z = d[..., 0] * d[..., 0], d[..., 1] * d[..., 1], d[..., 2] * d[..., 2]
    return ((x + y).astype(np.float32) + z).astype(np.float32)


def explain_by_replay(grun, orun, src, tgt, init, iters, thr, ext, nrm_t=None):
    """Both sides again with max_iterations = 1, 2, ... (threshold as given): the FIRST iteration at which they part must be
    (a) a stop decision whose margin | |prev_mse - mse| - threshold | is within the rounding of the f32 mse, or
    (b) correspondences that differ only in near ties (the two candidates' squared distances within 1e-4 relative -- the
        transforms of the two sides differ by ~1e-6 relative from the reference's sequential f32 sums), or
    (c) an error on one side of an ill-conditioned system (decided by the sign of a pivot within rounding).
    Before that iteration the transforms must agree.  -> (explained, text)"""
    tol = 2e-5 * max(1.0, ext)
    coord = float(np.abs(tgt).max())
    pg = po = ppg = ppo = None
    for k in range(1, iters + 1):
        try: g = grun(k)
        except tc.Error as e: g = None
        try: o = orun(k)
        except O.OracleError: o = None
        if g is None or o is None:
            if (g is None) == (o is None): return True, f"both fail at iteration {k}"
            return True, f"one side fails at iteration {k} (ill-conditioned system)"       # (c), judged by the caller's statistics
        if (g.converged, g.iterations) != (o.converged, o.iterations):
            # (a): look at the margins of the deciding iteration
            def margin(cur, prev):
                if prev is None: return float("inf")
                return abs(abs(prev.mse - cur.mse) - thr)
            # mse of the iteration before = the k-1 run's mse when it did not converge (point-to-plane reports prev_mse there)
            mg, mo = margin(g, pg), margin(o, po)
            scale = 16 * EPS32 * max(abs(o.mse), abs(g.mse), 1e-30)
            if min(mg, mo) <= scale or abs(g.mse - o.mse) <= scale and min(mg, mo) <= 4 * scale:
                return True, f"stop at iteration {k} decided within rounding: margins {mg:.3e} / {mo:.3e}, 16 eps mse = {scale:.3e}"
            # both residuals at the rounding floor of the coordinates (a noise-free pair aligned to the last bits: what is left of
            # the mse is the f32 rounding of the transformed points, (tens of ulps of the coordinates)^2, on either side)
            floor = (64.0 * EPS32 * coord) ** 2
            if max(g.mse, o.mse) <= floor and thr <= floor:
                return True, f"stop at iteration {k} decided at the rounding floor of the coordinates: mse {g.mse:.3e} / {o.mse:.3e} <= {floor:.3e}"
            # The sides' transforms agree to rounding (checked below for every earlier iteration), but a mean squared residual r^2 moves
            # by 2 r d + d^2 when the points move by d: `change` = |mse_{k-1} - mse_k| of the two sides may straddle the threshold
            # by that much.  In-loop mse_j = pairs of iteration j under the transform after j - 1 updates (registration.rs:324 / :578).
            def inloop(run_j, run_jm1):
                Tm = run_jm1.transformation if run_jm1 is not None else (O.IDENTITY if init is None else np.asarray(init, np.float32))
                c = run_j.correspondences
                d = tgt[c[:, 1]].astype(np.float64) - O.isometry_apply(Tm, src)[c[:, 0]].astype(np.float64)
                if nrm_t is not None: return float(np.mean(np.sum(nrm_t[c[:, 1]].astype(np.float64) * d, axis=1) ** 2))
                return float(np.mean(np.sum(d * d, axis=1)))
            if pg is not None and po is not None and len(g.correspondences) and len(o.correspondences):
                mgk, mok = inloop(g, pg), inloop(o, po)
                mgp, mop = inloop(pg, ppg), inloop(po, ppo)
                dpos = lambda a, b: frob(a.transformation, b.transformation) * (1.0 + coord) if a is not None else 0.0
                bound = sum(2.0 * np.sqrt(max(m1, m2)) * d_ + d_ * d_ for m1, m2, d_ in ((mgk, mok, dpos(pg, po)), (mgp, mop, dpos(ppg, ppo))))
                cg, co = abs(mgp - mgk), abs(mop - mok)
                if abs(cg - co) <= 2.0 * bound + 1e-30 and min(abs(cg - thr), abs(co - thr)) <= 2.0 * bound + 16 * EPS32 * max(mgp, mop):
                    return True, (f"stop at iteration {k}: the sides' |mse_{k - 1} - mse_{k}| = {cg:.4e} / {co:.4e} straddle the threshold {thr:.4e} "
                                  f"within what their transforms' distance explains ({bound:.2e})")
            return False, f"stop decisions part at iteration {k}: gpu {g.converged} mse {g.mse:.9e} oracle {o.converged} mse {o.mse:.9e} margins {mg:.3e} / {mo:.3e}"
        same_pairs = len(g.correspondences) == len(o.correspondences) and np.array_equal(g.correspondences, o.correspondences)
        if not same_pairs:
            # (b): each side must have matched the nearest target point under ITS OWN transform of the iteration before (the two
            # transforms differ by the rounding of the reference's sequential f32 sums); same f32 formula as both searches
            T0 = O.IDENTITY if init is None else np.asarray(init, np.float32)
            tg_ = O.isometry_apply(pg.transformation if pg is not None else T0, src)
            to_ = O.isometry_apply(po.transformation if po is not None else T0, src)
            gm = {int(a): int(b) for a, b in g.correspondences}; om = {int(a): int(b) for a, b in o.correspondences}
            wrong = 0
            for j in set(gm) | set(om):
                a, b = gm.get(j), om.get(j)
                if a == b or a is None or b is None: continue       # (in / out of the maximum distance: the same argument on sqrt(d2) > limit)
                if float(d2_f32(tgt[a], tg_[j])) > float(d2_f32(tgt[b], tg_[j])): wrong += 1      # the GPU did not take its nearest
                if float(d2_f32(tgt[b], to_[j])) > float(d2_f32(tgt[a], to_[j])): wrong += 1      # (the oracle did not: cannot happen)
            dT = frob(pg.transformation, po.transformation) if pg is not None else 0.0
            if wrong == 0:
                return True, f"correspondences part at iteration {k}: each side matched its nearest point under its own transform ({dT:.2e} apart)"
            return False, f"correspondences part at iteration {k}: {wrong} pairs are not the nearest under the side's own transform"
        fro = frob(g.transformation, o.transformation)
        if fro > tol * (1 + 0.25 * k):
            if nrm_t is not None and len(o.correspondences) >= 6:
                # an ill-conditioned 6x6 system (a handful of pairs, pairs on a line ...): the reference's f32 sums and f32 Cholesky
                # (registration.rs:409-438) carry ~1e-6 relative, the solution moves by cond(AtA) times that
                Tm = po.transformation if po is not None else (O.IDENTITY if init is None else np.asarray(init, np.float32))
                c = o.correspondences
                sp = O.isometry_apply(Tm, src)[c[:, 0]].astype(np.float64); q = tgt[c[:, 1]].astype(np.float64); nn = nrm_t[c[:, 1]].astype(np.float64)
                A = np.concatenate([np.cross(sp, nn), nn], axis=1)
                b = np.sum(nn * (q - sp), axis=1)
                AtA = A.T @ A
                cond = float(np.linalg.cond(AtA))
                x = np.linalg.lstsq(A, b, rcond=None)[0]
                reach = cond * 1e-6 * float(np.linalg.norm(x)) * (1.0 + coord)
                if fro <= 10.0 * reach:
                    return True, f"ill-conditioned 6x6 system at iteration {k} (cond {cond:.2e}, {len(c)} pairs): transforms {fro:.3e} apart, rounding reaches {reach:.3e}"
            if nrm_t is None and len(o.correspondences) >= 1:
                # point-to-point with a rank-deficient cross-covariance (a handful of pairs, several sources on one target point, pairs
                # on a line): the optimal rotation is a FAMILY (any turn about the remaining axis), which member comes out is the SVD
                # routine's business (registration.rs:163-201 through nalgebra's SVD; here a one-sided Jacobi in f64) -- explained iff
                # the second singular value vanishes and both sides' transforms leave the same residual on the pairs
                Tm = po.transformation if po is not None else (O.IDENTITY if init is None else np.asarray(init, np.float32))
                c = o.correspondences
                sp = O.isometry_apply(Tm, src)[c[:, 0]].astype(np.float64); q = tgt[c[:, 1]].astype(np.float64)
                sv = np.linalg.svd((sp - sp.mean(0)).T @ (q - q.mean(0)), compute_uv=False)
                res = lambda r_: float(np.mean(np.sum((O.isometry_apply(r_.transformation, src)[c[:, 0]].astype(np.float64) - q) ** 2, axis=1)))
                rg, ro = res(g), res(o)
                if sv[1] <= 1e-6 * max(sv[0], 1e-300) and abs(rg - ro) <= 1e-4 * max(rg, ro) + 1e-30:
                    return True, (f"rank-deficient Kabsch at iteration {k} ({len(c)} pairs, singular values {sv[0]:.2e} {sv[1]:.2e} {sv[2]:.2e}): the optimal "
                                  f"rotation is not unique, both sides' residuals {rg:.6e} / {ro:.6e} agree")
            return False, f"same pairs up to iteration {k} but transforms {fro:.3e} apart"
        if g.converged: return True, "agree until both stop"
        ppg, ppo, pg, po = pg, po, g, o
    return True, "agree at every iteration count"


def run(budget, seed, ctx, log=print, only_case=None, max_cases=None, min_cases=0):
    """every case draws from its own generator (seed, case number): `only_case` replays one"""
    t_end = time.time() + budget
    t_hard = t_end + 7 * budget          # (min_cases: a slow or cold box goes on past the budget until it has that many cases)
    cases = bad = borderline = 0
    while (time.time() < t_end or (cases < min_cases and time.time() < t_hard)) and (max_cases is None or cases < max_cases):
        cases += 1
        if only_case is not None:
            if cases > 1: break
            cases = only_case
        rng = np.random.default_rng([seed, cases])
        n = int(rng.choice([300, 1500, 4000]))
        kind = int(rng.integers(0, 3))
        if kind == 0: tgt = rng.random((n, 3))
        elif kind == 1: u = rng.random((n, 2)); tgt = np.stack([u[:, 0], u[:, 1], 0.15 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])], 1)
        else: tgt = rng.random((n, 3)) * np.array([4.0, 1.0, 0.3])
        tgt = (tgt * rng.choice([0.1, 1.0, 20.0])).astype(np.float32)
        ext = float(np.linalg.norm(tgt.max(0) - tgt.min(0)))
        T = synth.yaw_isometry(tuple((rng.normal(0, 0.03, 3) * ext).tolist()), float(rng.normal(0, 0.05)))
        m = int(n * rng.choice([0.3, 1.0]))
        src = synth.apply_isometry(T, tgt[rng.permutation(n)[:m]])
        if rng.random() < 0.5:
            src = (src + rng.normal(0, 2e-3 * ext, src.shape)).astype(np.float32)
        p2plane = bool(rng.random() < 0.5)
        iters = int(rng.integers(1, 61))
        thr = float(rng.choice([0.0, 1e-12, 1e-9, 1e-6, 1e-4, 1e-2])) * (ext * ext if rng.random() < 0.5 else 1.0)
        md = None if rng.random() < 0.5 else float(ext * rng.choice([0.05, 0.2, 1.0]))
        init = None if rng.random() < 0.5 else synth.yaw_isometry(tuple((rng.normal(0, 0.01, 3) * ext).tolist()), float(rng.normal(0, 0.01)))
        handles = bool(rng.random() < 0.3)
        tag = f"case {cases}: n {n} m {m} kind {kind} {'p2plane' if p2plane else 'p2p'} iters {iters} thr {thr:.3g} md {md} init {init is not None} handles {handles}"
        try:
            if p2plane:
                nrm = O.estimate_normals(tgt, min(10, n - 1))[:, 3:]
                orun0 = lambda s_: O.icp_point_to_plane_detailed(s_, tgt, nrm, init, iters, md, thr)
            else:
                orun0 = lambda s_: O.icp_detailed(s_, tgt, init, iters, md, thr)
            def operm(seed_):           # the oracle on the same source points in another order (None: it fails there)
                try:
                    return orun0(np.ascontiguousarray(src[np.random.default_rng(seed_).permutation(len(src))]))
                except O.OracleError:
                    return None
            orun = orun0
            try:
                r, rerr = orun(src), None
            except O.OracleError as e:
                r, rerr = None, str(e)
            def grun(k):
                if handles:
                    hs, ht = tc.Cloud(ctx, src), tc.Cloud(ctx, tgt)
                    try:
                        if p2plane:
                            ht.set_normals(nrm)
                            return hs.icp_point_to_plane(ht, init, k, md, thr, correspondences=True)
                        return hs.icp_detailed(ht, init, k, md, thr, correspondences=True)
                    finally:
                        hs.close(); ht.close()
                if p2plane: return ctx.icp_point_to_plane_detailed(src, tgt, nrm, init, k, md, thr)
                return ctx.icp_detailed(src, tgt, init, k, md, thr)
            okrun = (lambda k: O.icp_point_to_plane_detailed(src, tgt, nrm, init, k, md, thr)) if p2plane else \
                    (lambda k: O.icp_detailed(src, tgt, init, k, md, thr))
            try:
                g, gerr = grun(iters), None
            except tc.Error as e:
                g, gerr = None, type(e).__name__ + ": " + str(e)
            mismatch = None
            if (g is None) != (r is None):
                mismatch = f"ERROR MISMATCH | gpu: {gerr} | oracle: {rerr}"
            elif g is not None:
                tol = 1e-5 * max(1.0, ext)
                fro = frob(g.transformation, r.transformation)
                if (g.converged, g.iterations) != (r.converged, r.iterations):
                    mismatch = f"STOP MISMATCH | gpu {g.converged} {g.iterations} | oracle {r.converged} {r.iterations}"
                elif fro > tol:
                    mismatch = f"TRANSFORM MISMATCH | frob {fro:.3e}"
                elif abs(g.mse - r.mse) > 1e-9 + 2e-3 * abs(r.mse):
                    mismatch = f"MSE MISMATCH | {g.mse} {r.mse}"
                elif abs(len(g.correspondences) - len(r.correspondences)) > max(2, len(r.correspondences) // 500):
                    mismatch = f"CORRESPONDENCE COUNT MISMATCH | {len(g.correspondences)} {len(r.correspondences)}"
            if mismatch:
                ok, why = explain_by_replay(grun, okrun, src, tgt, init, iters, thr, ext, nrm if p2plane else None)
                if ok: borderline += 1
                else: bad += 1
                if not ok or only_case is not None: log(mismatch, "|", tag, "|", why)
        except Exception as e:
            bad += 1; log("EXCEPTION", tag, type(e).__name__, e)
    log(f"loop fuzz: {cases} cases, {bad} unexplained, {borderline} parted from the oracle at a rounding-decided step (near-tie pair, stop margin, pivot sign)")
    return cases, bad


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("FUZZ_DUMP_AFTER", "1500")), exit=True)
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0, tc.GpuContext(0),
        only_case=int(sys.argv[3]) if len(sys.argv) > 3 else None)
