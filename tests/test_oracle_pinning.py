"""Operation-level pinning of the oracle's restated nalgebra routines on the matrices the BENCHMARK produces (VERDICT r1
item 3a): not 10^4 random matrices but all 10^6 neighbourhood covariances of the 1 M-point cloud and the 6x6 / 3x3 systems of
its registration, each against LAPACK in f64.  What is asserted is backward stability at f32 precision -- the property
nalgebra's routines have -- as a function of the conditioning: eigenvector angle <= C eps / gap, solve error <= C eps cond.
The reports (worst eigenvector angle per eigen-gap decade) are printed with `pytest -s`."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import oracle as O
from threecrate_amd import synth

EPS = float(np.finfo(np.float32).eps)
N, K = 1_000_000, 16


@pytest.fixture(scope="module")
def bench_cloud():
    src, tgt, T = synth.registration_pair(N, seed=1, transform=synth.harness_transform(), noise_sigma=1e-4)
    return src, tgt


@pytest.fixture(scope="module")
def covariances(bench_cloud):
    """the 17-point covariance matrices normals.rs:164-177 hands to symmetric_eigen, in f32 and in the reference's order of
    operations (neighbours ascending, self last; centroid then sum of outer products, both sequential)"""
    _, tgt = bench_cloud
    d, idx = cKDTree(tgt).query(tgt, k=K + 1, workers=-1)
    self_col = idx == np.arange(len(tgt))[:, None]
    # neighbours without self (self is the first column except under exact duplicates), then self
    order = np.argsort(self_col, axis=1, kind="stable")
    idx = np.take_along_axis(idx, order, axis=1)
    P = tgt[idx]                                                   # (n, 17, 3) f32
    c = np.zeros((len(tgt), 3), np.float32)
    for j in range(K + 1):
        c = (c + P[:, j]).astype(np.float32)
    c = (c / np.float32(K + 1)).astype(np.float32)
    cov = np.zeros((len(tgt), 3, 3), np.float32)
    for j in range(K + 1):
        dlt = (P[:, j] - c).astype(np.float32)
        cov = (cov + dlt[:, :, None] * dlt[:, None, :]).astype(np.float32)
    return (cov / np.float32(K + 1)).astype(np.float32)


def test_symmetric_eigen_on_all_benchmark_covariances(covariances):
    cov = covariances
    ev, q = O.symmetric_eigen3_batch(cov)
    w, v = np.linalg.eigh(cov.astype(np.float64))                  # ascending
    # eigenvalues: absolute error relative to the matrix norm (= the largest eigenvalue)
    eval_err = np.abs(np.sort(ev, axis=1) - w).max(1) / (EPS * np.maximum(w[:, 2], 1e-300))
    print(f"\nworst eigenvalue error: {eval_err.max():.2f} eps * norm")
    assert eval_err.max() <= 16.0
    # orthonormal eigenvector matrices
    qtq = np.einsum("nij,nik->njk", q.astype(np.float64), q.astype(np.float64))
    assert np.abs(qtq - np.eye(3)).max() < 1e-5
    # the eigenvector normals.rs:186-194 takes (first strictly smallest eigenvalue) against LAPACK's, by relative gap
    imin = np.argmin(ev, axis=1)
    vmin = np.take_along_axis(q, imin[:, None, None].repeat(3, 1), axis=2)[:, :, 0].astype(np.float64)
    cosang = np.clip(np.abs((vmin * v[:, :, 0]).sum(1)), 0.0, 1.0)
    angle = np.arcsin(np.clip(np.linalg.norm(np.cross(vmin, v[:, :, 0]), axis=1), 0.0, 1.0))      # (arccos near 1 has no digits)
    gap = (w[:, 1] - w[:, 0]) / np.maximum(w[:, 2], 1e-300)
    report = []
    for lo, hi in [(1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1.1)]:
        m = (gap >= lo) & (gap < hi)
        if m.any():
            report.append((lo, hi, int(m.sum()), float(angle[m].max()), float((angle[m] * gap[m] / EPS).max())))
    print("\nrel eigen-gap decade | matrices | worst eigenvector angle [rad] | worst angle * gap / eps")
    for r in report:
        print(f"  [{r[0]:.0e}, {r[1]:.0e})  {r[2]:8d}   {r[3]:.3e}   {r[4]:.2f}")
    # Backward stability would mean angle <= C eps / gap (Davis-Kahan).  The restated nalgebra algorithm has it for the bulk of
    # the matrices and NOT for all of them: its final 2 x 2 block builds the rotation from (lambda - d, off) (symmetric_eigen.rs,
    # `basis`), a difference that cancels when the QR sweeps leave `off` just above the deflation threshold; the rotation
    # angle of that last step then is rounding noise and mixes the two columns it touches, whatever the eigen-gap.  The oracle
    # keeps that behaviour on purpose (the HIP kernel runs the same arithmetic and matches it bit for bit); this test pins HOW
    # OFTEN it matters on the benchmark's own matrices.  Whether threecrate itself lands on the same vectors for those few
    # points depends on nalgebra's exact instruction sequence in that step, which cannot be checked in this container:
    # that is the residual "parity unpinned" of DESIGN.md section 2.
    ok = gap >= 1e-3
    ratio = angle[ok] * gap[ok] / EPS
    q50, q99, q999 = np.quantile(ratio, [0.5, 0.99, 0.999])
    n_1e3 = int((angle[ok] > 1e-3).sum())
    n_budget = int(((1.0 - cosang[ok]) > 1e-4).sum())
    print(f"angle * gap / eps: median {q50:.2f}, 99 % {q99:.0f}, 99.9 % {q999:.0f}, worst {ratio.max():.0f}; "
          f"eigenvectors off by > 1e-3 rad: {n_1e3} of {int(ok.sum())}; beyond the 1e-4 cosine budget of the normals: {n_budget}")
    assert q50 <= 2.0 and q99 <= 1000.0
    assert n_1e3 <= 1e-4 * ok.sum() and n_budget <= 2e-5 * ok.sum()


def _system(vec29):
    A = np.zeros((6, 6))
    A[np.triu_indices(6)] = vec29[:21]
    A = A + A.T - np.diag(np.diag(A))
    return A, vec29[21:27].copy()


def test_solvers_on_the_registration_systems(bench_cloud):
    """the 6x6 normal equations of the benchmark registration at several transforms (the harness start, two steps in, the
    answer): cholesky6 / lu6 in f32 against numpy in f64, error bounded by the condition number"""
    src, tgt = bench_cloud
    m = 200_000                                                     # every 5th point: the same systems up to a factor
    src, tgt = np.ascontiguousarray(src[::5][:m]), np.ascontiguousarray(tgt[::5][:m])
    nrm = O.estimate_normals(tgt, K)[:, 3:]
    tree = O.KdTree(tgt)
    T0 = O.IDENTITY
    T2 = O.icp_point_to_plane_detailed(src, tgt, nrm, None, 2, None, 0.0).transformation
    Tf = synth.harness_transform()
    for T in (T0, T2, Tf):
        sums, _ = O.p2plane_partial(src, 0, len(src), tree, nrm, T)
        A, b = _system(sums)
        x64 = np.linalg.solve(A, b)
        cond = np.linalg.cond(A)
        for solve in (O.cholesky6_solve, O.lu6_solve):
            x = solve(A.astype(np.float32), b.astype(np.float32))
            assert x is not None
            rel = np.linalg.norm(x.astype(np.float64) - x64) / max(np.linalg.norm(x64), 1e-300)
            assert rel <= 8 * EPS * cond, (rel, cond)
        # point-to-point: the 3x3 cross-covariance of the same pairs -> svd3 and the rotation built from it
        s17, _ = O.p2p_partial(src, 0, len(src), tree, T)
        n = s17[16]
        ms, mq = s17[0:3] / n, s17[3:6] / n
        H = (s17[6:15].reshape(3, 3) - n * np.outer(ms, mq)).astype(np.float32)
        U, S, Vt = O.svd3(H)
        U64, S64, Vt64 = np.linalg.svd(H.astype(np.float64))
        assert np.abs(np.sort(S)[::-1] - S64).max() <= 16 * EPS * S64[0]
        assert np.abs(U.astype(np.float64) @ np.diag(S.astype(np.float64)) @ Vt.astype(np.float64) - H).max() <= 16 * EPS * S64[0]
        R = Vt.T.astype(np.float64) @ U.T.astype(np.float64)
        R64 = Vt64.T @ U64.T
        assert np.abs(R - R64).max() <= 64 * EPS * S64[0] / S64[2]          # well conditioned here: all three singular values large
        assert abs(np.linalg.det(R) - 1.0) < 1e-5
        qd = O.quat_from_matrix(R.astype(np.float32))
        x, y, z, wq = [float(v) for v in qd]
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * wq), 2 * (x * z + y * wq)],
                       [2 * (x * y + z * wq), 1 - 2 * (x * x + z * z), 2 * (y * z - x * wq)],
                       [2 * (x * z - y * wq), 2 * (y * z + x * wq), 1 - 2 * (x * x + y * y)]])
        assert np.abs(Rq - R).max() < 4e-6


def test_oracle_threads_follow_the_cpu_quota():
    """oracle.effective_cpus(): the OpenMP regions of the oracle run on the CPUs this process may actually use (scheduler affinity
    capped by the cgroup CPU quota), not on every CPU the box shows; tco_set_max_threads caps what `threads=0` means."""
    import os
    from oracle import oracle as O
    n = O.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert O.num_threads() <= n
    L = O.lib()
    try:
        L.tco_set_max_threads(1)
        assert O.num_threads() == 1
        pts = np.random.default_rng(0).random((2000, 3)).astype(np.float32)
        one = O.estimate_normals(pts, 8)
        L.tco_set_max_threads(n)
        assert np.array_equal(one, O.estimate_normals(pts, 8))        # the thread count never changes a result
    finally:
        L.tco_set_max_threads(n)
