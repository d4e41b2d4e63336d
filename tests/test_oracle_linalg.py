"""Cross-checks of the oracle's restated nalgebra / kd-tree pieces against LAPACK (numpy, f64),
scipy.spatial.cKDTree and brute force (how parity is established when the
reference cannot be built here)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import oracle as O
from threecrate_amd import synth


def test_symmetric_eigen_vs_lapack():
    rng = np.random.default_rng(0)
    worst_val, worst_vec = 0.0, 0.0
    for _ in range(10000):
        a = rng.standard_normal((3, 3)).astype(np.float32)
        a = ((a + a.T) / 2).astype(np.float32)
        ev, q = O.symmetric_eigen3(a)
        w, v = np.linalg.eigh(a.astype(np.float64))
        scale = np.abs(a).max()
        worst_val = max(worst_val, np.abs(np.sort(ev) - w).max() / scale)
        assert np.abs(q.T.astype(np.float64) @ q - np.eye(3)).max() < 5e-6       # orthonormal columns
        # eigenvector of the smallest eigenvalue when it is well separated
        if (w[1] - w[0]) > 0.2 * (w[2] - w[0]):
            i = int(np.argmin(ev))
            worst_vec = max(worst_vec, 1.0 - abs(float(q[:, i].astype(np.float64) @ v[:, 0])))
    assert worst_val < 5e-6, worst_val
    # the restated 2x2 deflation step (basis = (lambda - d, off)) loses digits when off is tiny:
    # observed eigenvector error up to ~1e-2 rad on unlucky matrices, i.e. 1-|cos| <~ 1e-4
    assert worst_vec < 2e-4, worst_vec


def test_symmetric_eigen_covariances():
    """17-point neighbourhood covariances (the matrices normals.rs:181 actually decomposes)."""
    rng = np.random.default_rng(1)
    dev = []
    for _ in range(3000):
        x = (rng.random((17, 3)) * [0.02, 0.02, rng.choice([0.02, 0.004, 0.0005])] + rng.random(3)).astype(np.float32)
        c = np.cov(x.T, bias=True).astype(np.float32)
        ev, q = O.symmetric_eigen3(c)
        w, v = np.linalg.eigh(c.astype(np.float64))
        assert np.abs(np.sort(ev) - w).max() <= 2e-6 * np.abs(c).max() + 1e-12
        if (w[1] - w[0]) > 0.05 * w[2]:
            dev.append(1.0 - abs(float(q[:, int(np.argmin(ev))].astype(np.float64) @ v[:, 0])))
    assert max(dev) < 1e-4


def test_special_matrices():
    ev, q = O.symmetric_eigen3(np.zeros((3, 3)))
    assert np.all(ev == 0) and np.array_equal(q, np.eye(3, dtype=np.float32))
    ev, q = O.symmetric_eigen3(np.diag([3.0, 1.0, 2.0]))
    assert np.allclose(ev, [3, 1, 2]) and np.allclose(np.abs(q), np.eye(3))
    ev, q = O.symmetric_eigen3(np.array([[1, 0.5, 0], [0.5, 2, 0], [0, 0, 0.0]]))
    i = int(np.argmin(ev))
    assert abs(ev[i]) < 1e-7 and abs(abs(q[2, i]) - 1) < 1e-6      # planar data -> normal = +-z


def test_svd_vs_lapack():
    rng = np.random.default_rng(2)
    for _ in range(10000):
        a = rng.standard_normal((3, 3)).astype(np.float32)
        u, s, vt = O.svd3(a)
        assert np.all(np.diff(s) <= 0) and np.all(s >= 0)                     # sorted descending, non-negative
        assert np.abs((u * s) @ vt - a).max() < 2e-5
        assert np.abs(s - np.linalg.svd(a.astype(np.float64), compute_uv=False)).max() < 1e-5
        assert np.abs(u.T @ u - np.eye(3)).max() < 1e-5 and np.abs(vt @ vt.T - np.eye(3)).max() < 1e-5


def test_kabsch_recovers_rotation():
    rng = np.random.default_rng(3)
    for _ in range(200):
        s = rng.random((50, 3)).astype(np.float32)
        ang = rng.uniform(-1, 1, 3)
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        T = np.array([q[0], q[1], q[2], q[3], *rng.uniform(-1, 1, 3)], np.float32)
        t = synth.apply_isometry(T, s)
        est = O.kabsch(s, t)
        assert np.linalg.norm(O.isometry_to_matrix(est).astype(np.float64) - synth.isometry_matrix(T)) < 2e-5


def test_cholesky_lu_vs_lapack():
    rng = np.random.default_rng(4)
    for _ in range(2000):
        m = rng.standard_normal((12, 6))
        a = (m.T @ m).astype(np.float32)
        b = rng.standard_normal(6).astype(np.float32)
        ref = np.linalg.solve(a.astype(np.float64), b.astype(np.float64))
        tol = 2e-5 * np.linalg.cond(a.astype(np.float64)) * max(1.0, np.abs(ref).max())
        x = O.cholesky6_solve(a, b)
        assert x is not None and np.abs(x - ref).max() < tol
        y = O.lu6_solve(a, b)
        assert y is not None and np.abs(y - ref).max() < tol
    # non-positive-definite -> Cholesky refuses (registration.rs:432-438 then falls back to LU)
    a = np.diag([1, 1, 1, 1, 1, -1]).astype(np.float32)
    assert O.cholesky6_solve(a, np.ones(6)) is None
    assert np.allclose(O.lu6_solve(a, np.ones(6)), [1, 1, 1, 1, 1, -1])
    assert O.lu6_solve(np.zeros((6, 6)), np.ones(6)) is None


def test_quat_from_matrix_roundtrip():
    rng = np.random.default_rng(5)
    for _ in range(500):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        T = np.array([q[0], q[1], q[2], q[3], 0, 0, 0], np.float32)
        R = synth.isometry_matrix(T)[:3, :3].astype(np.float32)
        qe = O.quat_from_matrix(R)
        Re = O.isometry_to_matrix(np.array([*qe, 0, 0, 0], np.float32))[:3, :3]
        assert np.abs(Re.astype(np.float64) - R).max() < 3e-6


def test_isometry_algebra_matches_f64():
    rng = np.random.default_rng(6)
    for _ in range(200):
        qa = rng.standard_normal(4); qa /= np.linalg.norm(qa)
        qb = rng.standard_normal(4); qb /= np.linalg.norm(qb)
        A = np.array([*qa, *rng.uniform(-2, 2, 3)], np.float32)
        B = np.array([*qb, *rng.uniform(-2, 2, 3)], np.float32)
        C = O.isometry_mul(A, B)
        assert np.abs(O.isometry_to_matrix(C).astype(np.float64) - synth.isometry_matrix(A) @ synth.isometry_matrix(B)).max() < 5e-6
        p = rng.uniform(-3, 3, (5, 3)).astype(np.float32)
        assert np.abs(O.isometry_apply(A, p) - synth.apply_isometry(A, p)).max() < 5e-6


def test_kdtree_knn_sets_vs_ckdtree_and_bruteforce():
    pts = synth.uniform_cloud(20000, seed=1)
    t = O.KdTree(pts)
    ck = cKDTree(pts.astype(np.float64))
    qs = synth.uniform_cloud(300, seed=9)
    for i in range(300):
        idx, d = t.find_k_nearest(qs[i], 17)
        _, ii = ck.query(qs[i].astype(np.float64), 17)
        assert set(idx.tolist()) == set(ii.tolist())
        bi, bd = O.brute_knn(pts, qs[i], 17)
        assert np.array_equal(np.sort(d), np.sort(bd))        # bit-identical distances (same f32 expression)
        assert np.all(np.diff(d) >= 0)


def test_knn_batch_parallel_equals_serial():
    pts = synth.uniform_cloud(5000, seed=2)
    i1, d1, c1 = O.knn_batch(pts, pts[:500], 11, threads=1)
    i8, d8, c8 = O.knn_batch(pts, pts[:500], 11, threads=0)
    assert np.array_equal(i1, i8) and np.array_equal(d1, d8) and np.array_equal(c1, c8)


def test_normals_vs_f64_pca():
    """Oracle normals against an independent float64 PCA over the same (exact) neighbour sets."""
    pts = synth.uniform_cloud(4000, seed=3)
    out = O.estimate_normals(pts, 16)
    ck = cKDTree(pts.astype(np.float64))
    _, nb = ck.query(pts.astype(np.float64), 17)
    bad = 0
    for i in range(0, 4000, 7):
        x = pts[nb[i]].astype(np.float64)
        w, v = np.linalg.eigh(np.cov(x.T, bias=True))
        if (w[1] - w[0]) < 0.02 * w[2]:
            continue
        if 1 - abs(float(out[i, 3:].astype(np.float64) @ v[:, 0])) > 1e-4:
            bad += 1
    assert bad == 0


def test_icp_recovers_known_transform_both_variants():
    src, tgt, T = synth.registration_pair(8000, seed=5)
    r = O.icp_detailed(src, tgt, None, 20, None, 0.0)
    assert np.linalg.norm(O.isometry_to_matrix(r.transformation).astype(np.float64) - synth.isometry_matrix(T)) < 1e-5
    n = O.estimate_normals(tgt, 16)[:, 3:]
    r = O.icp_point_to_plane_detailed(src, tgt, n, None, 20, None, 0.0)
    assert np.linalg.norm(O.isometry_to_matrix(r.transformation).astype(np.float64) - synth.isometry_matrix(T)) < 1e-5
    assert r.iterations == 20 and not r.converged


def test_partial_sums_reproduce_full_iteration():
    """tco_p2plane_partial over two shards == one pass over everything (the sharded-ICP contract)."""
    src, tgt, T = synth.registration_pair(3000, seed=6)
    n = O.estimate_normals(tgt, 10)[:, 3:]
    tree = O.KdTree(tgt)
    a, ca = O.p2plane_partial(src, 0, 1500, tree, n, O.IDENTITY)
    b, cb = O.p2plane_partial(src, 1500, 3000, tree, n, O.IDENTITY)
    full, cf = O.p2plane_partial(src, 0, 3000, tree, n, O.IDENTITY)
    assert np.allclose(a + b, full, rtol=1e-13, atol=1e-13)
    assert np.array_equal(np.concatenate([ca, cb]), cf)
    assert full[28] == 3000
