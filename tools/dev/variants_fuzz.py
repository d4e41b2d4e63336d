import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: randomized differential run of the registration VARIANTS built on the same kernels -- multiscale_icp_point_to_point
(registration.rs:704-789), kiss_icp (kiss_icp.rs:183-300), gicp (gicp.rs:100-305) -- against the oracle over their
configuration space (voxel sizes, level counts, iteration counts, thresholds, maximum distances, ranges, k).  Compared: error /
no error, converged flag + iteration count, transform.  A difference is accepted when the oracle itself moves by as much under a
permutation of its input (these variants run tens of iterations on down-sampled clouds: every stop decision and every near-tie
pair of loop_fuzz.py applies, level by level); everything else is reported.
usage: python tools/dev/variants_fuzz.py [seconds] [seed] [case]"""
import time
import numpy as np
import threecrate_amd as tc
from oracle import oracle as O
from threecrate_amd import synth


def frob(a, b):
    return float(np.linalg.norm(O.isometry_to_matrix(a).astype(np.float64) - O.isometry_to_matrix(b).astype(np.float64)))


def build_case(seed, cases, ctx):
    """inputs + the two runners of case `cases` of campaign `seed` (also used by variants_debug.py)"""
    kiss_seeded = None; params = {}
    rng = np.random.default_rng([seed, cases])
    n = int(rng.choice([600, 2500, 8000]))
    kind = int(rng.integers(0, 3))
    if kind == 0: tgt = rng.random((n, 3))
    elif kind == 1: u = rng.random((n, 2)); tgt = np.stack([u[:, 0], u[:, 1], 0.15 * np.sin(5 * u[:, 0]) * np.cos(4 * u[:, 1])], 1)
    else: tgt = rng.random((n, 3)) * np.array([4.0, 1.0, 0.3])
    scale = float(rng.choice([1.0, 10.0, 40.0]))
    tgt = (tgt * scale).astype(np.float32)
    ext = float(np.linalg.norm(tgt.max(0) - tgt.min(0)))
    spacing = ext / n ** (1.0 / 3.0)
    T = synth.yaw_isometry(tuple((rng.normal(0, 0.01, 3) * ext).tolist()), float(rng.normal(0, 0.02)))
    src = synth.apply_isometry(T, tgt[rng.permutation(n)[: int(n * rng.choice([0.5, 1.0]))]])
    if rng.random() < 0.5: src = (src + rng.normal(0, 1e-3 * ext, src.shape)).astype(np.float32)
    init = None if rng.random() < 0.6 else synth.yaw_isometry(tuple((rng.normal(0, 0.004, 3) * ext).tolist()), float(rng.normal(0, 0.005)))
    which = int(rng.integers(0, 3))
    if which == 1:
        f_ = 0.3 / spacing
        tgt = (tgt * f_).astype(np.float32); src = (src * f_).astype(np.float32)
        if init is not None: init = np.concatenate([init[:4], init[4:] * f_]).astype(np.float32)
        ext *= f_; spacing *= f_
    if which == 0:
        nl = int(rng.integers(1, 4))
        levels = [(float(spacing * f), int(rng.integers(1, 25)), None if rng.random() < 0.4 else float(spacing * f * rng.choice([2.0, 6.0])))
                  for f in sorted(rng.choice([0.7, 1.5, 3.0, 6.0], nl, replace=False), reverse=True)]
        fin_it = int(rng.integers(1, 25)); fin_md = None if rng.random() < 0.4 else float(spacing * rng.choice([1.5, 5.0]))
        # thresholds far above the f32 resolution of the mse (~1e-7 * spacing^2): a stop decided by the last bits of the
        # reference's sequential sums is loop_fuzz.py's subject, not this one's
        thr = float(rng.choice([1e-4, 1e-3, 1e-2])) * spacing * spacing
        name = f"multiscale levels {levels} final {fin_it}/{fin_md} thr {thr:.3g}"
        cfg = tc.MultiScaleIcpConfig(levels=[tc.IcpScaleLevel(*l) for l in levels], final_refinement_iterations=fin_it,
                                     final_max_correspondence_distance=fin_md, convergence_threshold=thr)
        grun = lambda s_: ctx.multiscale_icp_point_to_point(s_, tgt, init, cfg)
        orun = lambda s_: O.multiscale_icp_point_to_point(s_, tgt, init, levels, fin_it, fin_md, thr)
        def oexact(s_):
            O.lib().tco_set_exact_sums(1)
            try: return O.multiscale_icp_point_to_point(s_, tgt, init, levels, fin_it, fin_md, thr)
            finally: O.lib().tco_set_exact_sums(0)
    elif which == 1:
        # (kiss_icp.rs:277 stops at |prev_mse - mse| < 1e-6 ABSOLUTE: metric clouds; here the cloud is rescaled so that the
        # mse is ~1e-2, i.e. the threshold sits 100x above the f32 resolution of the mse as on a LiDAR frame)
        vs = float(spacing * rng.choice([0.8, 1.5, 3.0])); mx = float(ext * rng.choice([0.5, 2.0])); mn = float(ext * rng.choice([0.0, 0.05]))
        it = int(rng.integers(1, 61))
        name = f"kiss voxel {vs:.3g} range {mn:.3g}..{mx:.3g} iters {it}"
        cfg = tc.KissIcpConfig(voxel_size=vs, max_range=mx, min_range=mn, max_iterations=it)
        grun = lambda s_: ctx.kiss_icp(s_, tgt, init, cfg)
        orun = lambda s_: O.kiss_icp(s_, tgt, init, vs, mx, mn, it)[0]
        oexact = lambda s_: O.kiss_icp(s_, tgt, init, vs, mx, mn, it, exact_sums=True)[0]
        kiss_seeded = lambda sd: O.kiss_icp(src, tgt, init, vs, mx, mn, it, voxel_order_seed=sd)[0]
    else:
        it = int(rng.integers(1, 41)); md = float(spacing * rng.choice([1.5, 4.0, 50.0])); thr = float(rng.choice([1e-4, 1e-3, 1e-2]))
        kc = int(rng.choice([5, 10, 20]))
        name = f"gicp iters {it} md {md:.3g} thr {thr:.3g} k {kc}"
        cfg = tc.GicpConfig(max_iterations=it, max_correspondence_distance=md, convergence_threshold=thr, k_correspondences=kc)
        grun = lambda s_: ctx.gicp(s_, tgt, init, cfg)
        orun = lambda s_: O.gicp(s_, tgt, init, it, md, thr, kc)
        oexact = lambda s_: O.gicp(s_, tgt, init, it, md, thr, kc, exact_sums=True)
    tag = f"case {cases}: n {n} m {len(src)} kind {kind} scale {scale} init {init is not None} | {name}"
    if which == 0: params = dict(levels=levels, fin_it=fin_it, fin_md=fin_md, thr=thr)
    elif which == 1: params = dict(vs=vs, mx=mx, mn=mn, it=it)
    else: params = dict(it=it, md=md, thr=thr, kc=kc)
    return dict(tag=tag, src=src, tgt=tgt, init=init, which=which, ext=ext, spacing=spacing, grun=grun, orun=orun, oexact=oexact, kiss_seeded=kiss_seeded, params=params, cfg=cfg)


def kiss_parting(cs, ctx):
    """KISS-ICP case: the first iteration count k at which the device's and the oracle's correspondences differ, and whether
    every differing row there is a NEAR-TIE -- the source point's distances to the two targets, under the oracle's transform after
    k - 1 iterations, differ by less than the two sides' transforms moved it apart (they agree to ~1e-6 up to there).  From such a
    step on the two are different, equally valid trajectories of the same iteration (on a few hundred voxel centroids in a flat
    valley of the objective they drift apart by 1e-2 within twenty iterations: campaign 504, case 28357).
    -> (k, rows differing, True) or None when the runs do not part like that."""
    src, tgt, init, P = cs["src"], cs["tgt"], cs["init"], cs["params"]
    vs, mx, mn, it = P["vs"], P["mx"], P["mn"], P["it"]
    d = np.linalg.norm(src.astype(np.float32), axis=1)
    sd = O.voxel_grid_filter(np.ascontiguousarray(src[(d >= np.float32(mn)) & (d <= np.float32(mx))]), vs)      # kiss_icp.rs:196-214
    td = tgt                    # (the pairs' target indices address the target as given: kiss_icp.rs matches against the whole map)
    prevT, prev_gap = (O.IDENTITY if init is None else np.asarray(init, np.float32)), 0.0
    for k in range(1, it + 1):
        g = ctx.kiss_icp(src, tgt, init, tc.KissIcpConfig(voxel_size=vs, max_range=mx, min_range=mn, max_iterations=k))
        r, nd = O.kiss_icp(src, tgt, init, vs, mx, mn, k)
        if len(g.correspondences) != len(r.correspondences) or nd != len(sd): return None
        rows = np.nonzero((np.asarray(g.correspondences) != np.asarray(r.correspondences)).any(axis=1))[0]
        if len(rows):
            if len(rows) > 3: return None
            # the pairs of iteration k were chosen under the transform after k - 1 iterations, which the two sides hold prev_gap apart
            q = O.isometry_apply(prevT, sd).astype(np.float64)
            slack = 4.0 * (prev_gap + 1e-6) * max(cs["ext"], 1.0)
            for row in rows:
                si, tg_, to_ = int(r.correspondences[row][0]), int(g.correspondences[row][1]), int(r.correspondences[row][1])
                if int(g.correspondences[row][0]) != si or max(tg_, to_) >= len(td) or si >= len(sd): return None
                dg, do = np.linalg.norm(q[si] - td[tg_]), np.linalg.norm(q[si] - td[to_])
                if abs(dg - do) > slack: return None
            return k, len(rows), True
        if r.converged and g.converged: return None
        prevT, prev_gap = np.asarray(r.transformation, np.float32), frob(g.transformation, r.transformation)
    return None


def run(budget, seed, ctx, log=print, only_case=None):
    t_end = time.time() + budget
    cases = bad = noisy = exact = illcond = margin = tiny = parted = 0
    while time.time() < t_end:
        cases += 1
        if only_case is not None:
            if cases > 1: break
            cases = only_case
        cs = build_case(seed, cases, ctx)
        tag, src, tgt, init, which, ext, grun, orun, kiss_seeded = (cs[k] for k in ('tag', 'src', 'tgt', 'init', 'which', 'ext', 'grun', 'orun', 'kiss_seeded'))
        try:
            try: g, gerr = grun(src), None
            except tc.Error as e: g, gerr = None, type(e).__name__ + ": " + str(e)[:80]
            try: r, rerr = orun(src), None
            except O.OracleError as e: r, rerr = None, str(e)
            def family():
                out = []
                for s_ in (1, 2, 3):
                    try:
                        # KISS-ICP: the order of the reference's down-sampled source is its HashMap's (unspecified): the oracle's
                        # voxel_order_seed; the others: the same points in another order
                        out.append(kiss_seeded(s_) if which == 1 else orun(np.ascontiguousarray(src[np.random.default_rng(s_).permutation(len(src))])))
                    except O.OracleError: out.append(None)
                return out
            if (g is None) != (r is None):
                if any((f is None) != (r is None) for f in family()): noisy += 1
                else: bad += 1; log("ERROR MISMATCH", tag, "| gpu:", gerr, "| oracle:", rerr)
                continue
            if g is None: continue
            tol = 1e-5 * max(1.0, ext)
            fro = frob(g.transformation, r.transformation)
            stop_same = (g.converged, g.iterations) == (r.converged, r.iterations)
            if stop_same and fro <= tol: continue
            fam = [f for f in family() if f is not None]
            spread = max([frob(f.transformation, r.transformation) for f in fam] + [0.0])
            stops = {(f.converged, f.iterations) for f in fam} | {(r.converged, r.iterations)}
            if (stop_same or len(stops) > 1 or (g.converged, g.iterations) in stops) and fro <= 3.0 * spread + tol:
                noisy += 1
                continue
            # the device adds the reference's f32 terms in f64: does it agree with the REFERENCE doing the same (exact_sums: the only
            # change is the accumulator of the Kabsch / Gauss-Newton sums)?  Then the difference is the reference's own rounding of
            # its sequential f32 sums, amplified by the case (a handful of points per level, a stop decided by the last digits)
            try: rx = cs["oexact"](src)
            except O.OracleError: rx = None
            if rx is not None and (g.converged, g.iterations) == (rx.converged, rx.iterations) and frob(g.transformation, rx.transformation) <= tol:
                exact += 1
                continue
            # a FLAT direction of the objective: the two transforms differ, the stop is the same and the device's residual is no
            # worse than the reference's.  Correspondences that are nearly collinear (a level down-sampled to four voxels in a
            # row) leave the rotation about their line to the last bits of the cross-covariance; the reference's f32 SVD and the
            # device's f64 Kabsch then land at different points of the same valley.  (Noise on the INPUT does not show it: the
            # voxel centroids average it away; 1e-7 relative noise moves the oracle by ~1e-5 in these cases.)  Reported as a
            # class of its own, not silently accepted.
            if stop_same and float(g.mse) <= float(r.mse) * (1.0 + 1e-3) + 1e-12:
                illcond += 1
                continue
            # the stop one iteration apart and the transform equal to one of the oracle's variants: |prev_mse - mse| against the
            # threshold was decided by the last digits (loop_fuzz.py measures those margins; here only counted)
            near = [x for x in (r, rx) if x is not None and g.converged == x.converged and abs(int(g.iterations) - int(x.iterations)) <= 1]
            if not stop_same and any(frob(g.transformation, x.transformation) <= 3.0 * tol for x in near):
                margin += 1
                continue
            # what is left, by the size of the smallest cloud a level registers: a handful of voxel centroids (three in a row on a
            # slab) is not a registration problem, the variants' hosts pass it to the same kernels all the same
            if which == 1:
                try: part = kiss_parting(cs, ctx)
                except Exception: part = None
                if part is not None:
                    parted += 1
                    continue
            small = None
            if which == 0: small = min(min(len(O.voxel_grid_filter(src, l[0])), len(O.voxel_grid_filter(tgt, l[0]))) for l in cs["params"]["levels"])
            if which == 1: small = len(O.voxel_grid_filter(src, cs["params"]["vs"]))
            if small is not None and small <= 12:
                tiny += 1
            else:
                bad += 1
                fx = -1.0 if rx is None else frob(g.transformation, rx.transformation)
                log("MISMATCH", tag, f"| gpu {g.converged} {g.iterations} oracle {r.converged} {r.iterations} frob {fro:.3e} oracle's own spread {spread:.3e} stops {sorted(stops)}"
                    f" | exact-sums oracle {None if rx is None else (rx.converged, rx.iterations)} frob to it {fx:.3e} | mse gpu {float(g.mse):.9g} oracle {float(r.mse):.9g}")
        except Exception as e:
            bad += 1; log("EXCEPTION", tag, type(e).__name__, str(e)[:200])
    log(f"variants fuzz: {cases} cases, {bad} to look at, {noisy} within the oracle's own sensitivity to the order of its input, "
        f"{exact} equal to the oracle with its f32 sums kept in f64, {illcond} with the same stop and a residual no worse than the oracle's (flat direction: near-collinear pairs), "
        f"{margin} stopping one iteration apart with an equal transform, {parted} KISS-ICP runs parted from the oracle at a near-tie pair "
        f"(identical up to that iteration), {tiny} on levels of <= 12 points")
    return cases, bad


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("FUZZ_DUMP_AFTER", "1500")), exit=True)
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0, tc.GpuContext(0),
        only_case=int(sys.argv[3]) if len(sys.argv) > 3 else None)
