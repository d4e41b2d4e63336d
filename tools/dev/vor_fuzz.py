import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
"""dev: A/B of the inscribed-ball test on odd clouds over MANY iterations: the same seeded sequence of registrations is run in two
processes -- with the bounds from the first iteration on (TC_VOR_AFTER=1) and without them (TC_DEBUG=4) -- and must give the same
bits (transform, mse, iterations, correspondences): the test only ever skips searches whose answer is already known.
usage: python tools/dev/vor_fuzz.py [seconds per side] [seed]          tests/test_gpu_fuzz.py calls compare() with a short budget."""
import hashlib
import json
import subprocess
import time

import numpy as np


def child(budget, seed, min_cases=0):
    import threecrate_amd as tc
    from threecrate_amd import synth
    ctx = tc.GpuContext(0)
    rng = np.random.default_rng(seed)
    out = []
    t_end = time.time() + budget
    t_hard = t_end + 7 * budget          # (min_cases: a slow or cold box goes on past the budget until it has that many cases)
    case = 0
    while (time.time() < t_end or (case < min_cases and time.time() < t_hard)) and case < 400:
        case += 1
        kind = int(rng.integers(0, 5)); n = int(rng.choice([300, 2000, 9000, 40000, 150000]))
        if kind == 0: p = rng.random((n, 3))
        elif kind == 1: p = rng.random((n, 3)) * np.array([10.0, 3.0, 0.2])
        elif kind == 2: u = rng.random((n, 2)); p = np.stack([u[:, 0], u[:, 1], 0.1 * np.sin(6 * u[:, 0]) + 1e-4 * rng.normal(size=n)], 1)
        elif kind == 3: p = np.round(rng.random((n, 3)) * 16) / 16 + 1e-5 * rng.normal(size=(n, 3))
        else: p = rng.random((n, 3)); p[: n // 8] = p[0]      # exact duplicates (up to 18 750 in one cell: ranked deterministically up to 65 536)
        tgt = (p * rng.choice([1e-2, 1.0, 50.0])).astype(np.float32)
        ext = float(np.linalg.norm(tgt.max(0) - tgt.min(0))) + 1e-6
        spacing = ext / max(n, 2) ** (1 / 3)
        T = synth.yaw_isometry(tuple((rng.normal(0, 1.5, 3) * spacing).tolist()), float(rng.normal(0, 2.0) * spacing / ext))
        src = synth.apply_isometry(T, tgt[rng.permutation(n)[: max(8, int(n * rng.uniform(0.3, 1.0)))]])
        src = (src + rng.normal(0, rng.choice([0.0, 0.05, 0.3]) * spacing, src.shape)).astype(np.float32)
        if rng.random() < 0.2:
            src[rng.integers(0, len(src), 3)] = np.nan
            tgt[rng.integers(0, n, 3), rng.integers(0, 3, 3)] = np.inf
        md = None if rng.random() < 0.6 else float(spacing * rng.choice([1.0, 4.0, 30.0]))
        iters = int(rng.choice([3, 9, 25]))
        rec = {"case": case, "kind": kind, "n": n, "iters": iters}
        try:
            if rng.random() < 0.5:
                r = ctx.icp_detailed(src, tgt, None, iters, md, 0.0)
            else:
                nrm = ctx.estimate_normals(tgt, 10)
                if rng.random() < 0.5:
                    r = ctx.icp_point_to_plane_detailed(src, tgt, nrm, None, iters, md, 0.0)
                else:
                    t, s = tc.Cloud(ctx, tgt), tc.Cloud(ctx, src)
                    t.set_normals(nrm)
                    r = s.icp_point_to_plane(t, None, iters, md, 0.0, correspondences=True)
                    t.close(); s.close()
            rec.update(T=r.transformation.tobytes().hex(), mse=float(r.mse), it=int(r.iterations), conv=bool(r.converged),
                       corr=hashlib.sha1(np.ascontiguousarray(r.correspondences).tobytes()).hexdigest())
        except tc.Error as e:
            rec.update(err=type(e).__name__)
        out.append(rec)
    print("RESULT " + json.dumps(out))


DEV_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "threecrate_amd", "variants", "libthreecrate_hip_dev.so")


def compare(budget, seed, log=print, min_cases=0, what="ball"):
    """what = "ball": the inscribed-ball test from the first iteration on against none; "second": the second-neighbour certificate
    (default build) against TC_DEBUG=4096 (off)"""
    me = os.path.abspath(__file__)
    res = {}
    # the switches that change the road live in the development build only (csrc/Makefile `dev`); unless the caller names a library
    # (TC_HIP_LIB = a variant under test) the arm that needs a switch loads that build
    dev = {} if os.environ.get("TC_HIP_LIB") else {"TC_HIP_LIB": DEV_LIB}
    envs = (("with", {"TC_VOR_AFTER": "1"}), ("without", dict(dev, TC_DEBUG="4"))) if what == "ball" else (("with", {}), ("without", dict(dev, TC_DEBUG="4096")))
    for name, env in envs:
        p = subprocess.run([sys.executable, me, "--child", str(budget), str(seed), str(min_cases)], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        assert line, p.stderr[-3000:]
        res[name] = json.loads(line[0][7:])
    n = min(len(res["with"]), len(res["without"]))
    bad = 0
    for a, b in zip(res["with"][:n], res["without"][:n]):
        if a != b:
            bad += 1
            log("MISMATCH", a, b)
    log(f"{'inscribed-ball' if what == 'ball' else 'second-neighbour certificate'} A/B: {n} registrations compared, {bad} differ")
    return n, bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        n, bad = compare(float(sys.argv[1]) if len(sys.argv) > 1 else 40.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                         what=sys.argv[3] if len(sys.argv) > 3 else "ball")
        sys.exit(1 if bad else 0)
