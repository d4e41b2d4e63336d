"""ctypes loader + numpy wrappers for the CPU oracle (oracle/tc_oracle.c).

TEST INFRASTRUCTURE ONLY: the product package threecrate_amd never imports this module.
The oracle restates threecrate-algorithms' CPU path (nearest_neighbor.rs, normals.rs,
registration.rs, filtering.rs); see tc_oracle.h for citations and parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, INVALID_DATA, ALGORITHM = 0, 1, 2


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        super().__init__({1: "InvalidData", 2: "Algorithm"}.get(code, str(code)))


class _IcpResult(C.Structure):
    _fields_ = [("transform", C.c_float * 7), ("mse", C.c_float), ("iterations", C.c_uint64),
                ("converged", C.c_int32), ("n_corr", C.c_uint64),
                ("corr_src", C.POINTER(C.c_uint64)), ("corr_tgt", C.POINTER(C.c_uint64))]


def build(force=False):
    if os.environ.get("TC_ORACLE_LIB"):          # e.g. the ASan / UBSan build (oracle/Makefile `asan`, tools/sanitize_cpu.sh)
        return os.environ["TC_ORACLE_LIB"]
    so = os.path.join(_HERE, "libtc_oracle.so")
    src = os.path.join(_HERE, "tc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtc_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        try:
            _LIB = C.CDLL(so)
        except OSError:
            _LIB = C.CDLL(build(force=True))
        L = _LIB
        f32p, u64p, u32p, f64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_double))
        L.tco_kdtree_new.restype = C.c_void_p
        L.tco_kdtree_new.argtypes = [f32p, C.c_size_t]
        L.tco_kdtree_free.argtypes = [C.c_void_p]
        L.tco_kdtree_knn.restype = C.c_size_t
        L.tco_kdtree_knn.argtypes = [C.c_void_p, f32p, C.c_size_t, u64p, f32p]
        L.tco_kdtree_radius.restype = C.c_size_t
        L.tco_kdtree_radius.argtypes = [C.c_void_p, f32p, C.c_float, u64p, f32p, C.c_size_t]
        L.tco_brute_knn.restype = C.c_size_t
        L.tco_brute_knn.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, u64p, f32p]
        L.tco_knn_batch.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, C.c_size_t, u64p, f32p, u32p, C.c_int]
        L.tco_estimate_normals.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_float, C.c_int, C.c_int, f32p, f32p, C.c_int]
        L.tco_icp_point_to_point.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_size_t, C.c_float,
                                             C.c_float, C.POINTER(_IcpResult), C.c_int]
        L.tco_icp_point_to_point_checked.argtypes = L.tco_icp_point_to_point.argtypes
        L.tco_icp_point_to_plane.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_size_t,
                                             C.c_float, C.c_float, C.POINTER(_IcpResult), C.c_int]
        L.tco_icp.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_int]
        L.tco_icp.restype = None
        L.tco_voxel_grid_filter.argtypes = [f32p, C.c_size_t, C.c_float, f32p, C.POINTER(C.c_size_t)]
        L.tco_gicp_covariances.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_int]
        L.tco_gicp_covariances.restype = None
        L.tco_gicp.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_size_t, C.c_float, C.c_float, C.c_size_t,
                               C.POINTER(_IcpResult), C.c_int]
        L.tco_kiss_adaptive_threshold.argtypes = [f32p, C.c_float]
        L.tco_kiss_adaptive_threshold.restype = C.c_float
        L.tco_kiss_icp.argtypes = [f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_float, C.c_float, C.c_float, C.c_size_t,
                                   C.POINTER(_IcpResult), C.POINTER(C.c_size_t), C.c_int]
        L.tco_set_exact_sums.argtypes = [C.c_int]
        L.tco_set_exact_sums.restype = None
        L.tco_set_voxel_order_seed.argtypes = [C.c_uint64]
        L.tco_set_voxel_order_seed.restype = None
        L.tco_p2plane_partial.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_void_p, f32p, f32p, f32p, C.c_float, f64p, u32p]
        L.tco_p2p_partial.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_void_p, f32p, f32p, C.c_float, f64p, u32p]
        L.tco_symmetric_eigen3.argtypes = [f32p, f32p, f32p]
        L.tco_symmetric_eigen3_batch.argtypes = [f32p, C.c_size_t, f32p, f32p]
        L.tco_symmetric_eigen3_batch.restype = None
        L.tco_svd3.argtypes = [f32p, f32p, f32p, f32p]
        L.tco_cholesky6_solve.argtypes = [f32p, f32p, f32p]
        L.tco_lu6_solve.argtypes = [f32p, f32p, f32p]
        L.tco_quat_from_matrix.argtypes = [f32p, f32p]
        L.tco_kabsch.argtypes = [f32p, f32p, C.c_size_t, f32p, C.POINTER(C.c_int)]
        L.tco_isometry_apply.argtypes = [f32p, f32p, f32p]
        L.tco_isometry_mul.argtypes = [f32p, f32p, f32p]
        L.tco_isometry_to_matrix.argtypes = [f32p, f32p]
        L.tco_num_threads.restype = C.c_int
        L.tco_set_max_threads.argtypes = [C.c_int]
        L.tco_set_max_threads.restype = None
        L.tco_set_max_threads(effective_cpus())
    return _LIB


def effective_cpus():
    """CPUs this process can actually run on: the scheduler affinity capped by the container's CPU bandwidth quota (cgroup v2
    cpu.max / v1 cfs quota).  A GPU box shows 256 CPUs under a 16-core quota: 256 runnable OpenMP threads there are throttled,
    not run (a 60-iteration ICP on 1500 points takes seconds instead of milliseconds)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def _f32(a, shape_last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float32)


def num_threads():
    return lib().tco_num_threads()


class KdTree:
    """nearest_neighbor.rs:29-299"""

    def __init__(self, points):
        self.points = _f32(points, 3)
        self._h = lib().tco_kdtree_new(_p(self.points), len(self.points))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tco_kdtree_free(self._h)
            self._h = None

    def find_k_nearest(self, query, k):
        q = _f32(query).reshape(3)
        cap = max(1, min(k, max(1, len(self.points))))
        idx = np.zeros(cap, np.uint64)
        dist = np.zeros(cap, np.float32)
        m = lib().tco_kdtree_knn(self._h, _p(q), k, _p(idx, C.c_uint64), _p(dist))
        return idx[:m].copy(), dist[:m].copy()

    def find_radius_neighbors(self, query, radius):
        q = _f32(query).reshape(3)
        cap = max(1, len(self.points))
        idx = np.zeros(cap, np.uint64)
        dist = np.zeros(cap, np.float32)
        m = lib().tco_kdtree_radius(self._h, _p(q), radius, _p(idx, C.c_uint64), _p(dist), cap)
        return idx[:m].copy(), dist[:m].copy()


def brute_knn(points, query, k):
    pts = _f32(points, 3)
    q = _f32(query).reshape(3)
    cap = max(1, min(k, max(1, len(pts))))
    idx = np.zeros(cap, np.uint64)
    dist = np.zeros(cap, np.float32)
    m = lib().tco_brute_knn(_p(pts), len(pts), _p(q), k, _p(idx, C.c_uint64), _p(dist))
    return idx[:m].copy(), dist[:m].copy()


def knn_batch(points, queries, k, threads=0):
    pts, qs = _f32(points, 3), _f32(queries, 3)
    idx = np.zeros((len(qs), k), np.uint64)
    dist = np.zeros((len(qs), k), np.float32)
    cnt = np.zeros(len(qs), np.uint32)
    lib().tco_knn_batch(_p(pts), len(pts), _p(qs), len(qs), k, _p(idx, C.c_uint64), _p(dist), _p(cnt, C.c_uint32), threads)
    return idx, dist, cnt


def estimate_normals(points, k=10, radius=None, consistent_orientation=True, viewpoint=None, threads=0):
    """estimate_normals_with_config (normals.rs:257-357). Returns (n,6) [position, normal]."""
    pts = _f32(points, 3)
    out = np.zeros((len(pts), 6), np.float32)
    vp = None if viewpoint is None else _f32(viewpoint).reshape(3)
    rc = lib().tco_estimate_normals(_p(pts), len(pts), k, 0.0 if radius is None else float(radius),
                                    0 if radius is None else 1, 1 if consistent_orientation else 0,
                                    None if vp is None else _p(vp), _p(out), threads)
    if rc:
        raise OracleError(rc)
    return out


def estimate_normals_radius(points, radius, consistent_orientation, threads=0):
    """normals.rs:368-380 (k_neighbors fallback = 10)"""
    return estimate_normals(points, 10, radius, consistent_orientation, None, threads)


class IcpResult:
    def __init__(self, r, ns, cs, ct):
        self.transformation = np.array(list(r.transform), np.float32)
        self.mse = float(r.mse)
        self.iterations = int(r.iterations)
        self.converged = bool(r.converged)
        n = int(r.n_corr)
        self.correspondences = np.stack([cs[:n], ct[:n]], axis=1).astype(np.int64)

    @property
    def matrix(self):
        return isometry_to_matrix(self.transformation)


def _icp_common(fn, src, tgt, extra_pre, init, max_iters, tail, threads):
    s, t = _f32(src, 3), _f32(tgt, 3)
    r = _IcpResult()
    cs = np.zeros(max(1, len(s)), np.uint64)
    ct = np.zeros(max(1, len(s)), np.uint64)
    r.corr_src = _p(cs, C.c_uint64)
    r.corr_tgt = _p(ct, C.c_uint64)
    i7 = _f32(IDENTITY if init is None else init).reshape(7)
    rc = fn(_p(s), len(s), _p(t), len(t), *extra_pre, _p(i7), max_iters, *tail, C.byref(r), threads)
    if rc:
        raise OracleError(rc)
    return IcpResult(r, len(s), cs, ct)


def icp_detailed(src, tgt, init, max_iters, max_correspondence_distance=None, convergence_threshold=1e-6, threads=0, exact_sums=False):
    """registration.rs:258-370.  exact_sums (diagnostic): the Kabsch sums' f32 terms are added in f64 instead of the reference's
    sequential f32 (tc_oracle.c, tco_set_exact_sums)."""
    md = -1.0 if max_correspondence_distance is None else float(max_correspondence_distance)
    lib().tco_set_exact_sums(1 if exact_sums else 0)
    try:
        return _icp_common(lib().tco_icp_point_to_point, src, tgt, (), init, max_iters,
                           (C.c_float(md), C.c_float(convergence_threshold)), threads)
    finally:
        lib().tco_set_exact_sums(0)


def icp_point_to_point(src, tgt, init, max_iterations, convergence_threshold=1e-6, max_correspondence_distance=None, threads=0):
    """registration.rs:644-680"""
    md = -1.0 if max_correspondence_distance is None else float(max_correspondence_distance)
    return _icp_common(lib().tco_icp_point_to_point_checked, src, tgt, (), init, max_iterations,
                       (C.c_float(convergence_threshold), C.c_float(md)), threads)


def icp(src, tgt, init, max_iters, threads=0):
    """registration.rs:232-242: returns the 7-float isometry; errors -> init"""
    s, t = _f32(src, 3), _f32(tgt, 3)
    i7 = _f32(IDENTITY if init is None else init).reshape(7)
    out = np.zeros(7, np.float32)
    lib().tco_icp(_p(s), len(s), _p(t), len(t), _p(i7), max_iters, _p(out), threads)
    return out


def icp_point_to_plane_detailed(src, tgt, tgt_normals, init, max_iters, max_correspondence_distance=None,
                                convergence_threshold=1e-6, threads=0, exact_sums=False):
    """registration.rs:508-602.  exact_sums (diagnostic): the 6x6 system's per-pair f32 terms are added in f64 instead of the
    reference's sequential f32 (tc_oracle.c, tco_set_exact_sums)."""
    n = _f32(tgt_normals, 3)
    md = -1.0 if max_correspondence_distance is None else float(max_correspondence_distance)
    lib().tco_set_exact_sums(1 if exact_sums else 0)
    try:
        return _icp_common(lib().tco_icp_point_to_plane, src, tgt, (_p(n), len(n)), init, max_iters,
                           (C.c_float(md), C.c_float(convergence_threshold)), threads)
    finally:
        lib().tco_set_exact_sums(0)


def icp_point_to_plane(src, tgt, tgt_normals, init, max_iters, threads=0):
    """registration.rs:488-496"""
    return icp_point_to_plane_detailed(src, tgt, tgt_normals, init, max_iters, None, 1e-6, threads)


def gicp(src, tgt, init=None, max_iterations=50, max_correspondence_distance=1.0, convergence_threshold=1e-6,
         k_correspondences=20, threads=0, exact_sums=False):
    """gicp.rs:100-305 (GicpConfig defaults :31-40).  exact_sums (diagnostic): the Gauss-Newton system's f32 terms added in f64."""
    lib().tco_set_exact_sums(1 if exact_sums else 0)
    try:
        return _icp_common(lib().tco_gicp, src, tgt, (), init, max_iterations,
                           (C.c_float(max_correspondence_distance), C.c_float(convergence_threshold), C.c_size_t(k_correspondences)), threads)
    finally:
        lib().tco_set_exact_sums(0)


def gicp_covariances(points, k=20, threads=0):
    """compute_covariances (gicp.rs:52-86) -> (n, 3, 3) float32"""
    pts = _f32(points, 3)
    out = np.zeros((len(pts), 9), np.float32)
    lib().tco_gicp_covariances(_p(pts), len(pts), k, _p(out), threads)
    return out.reshape(-1, 3, 3)


def kiss_icp(src, tgt, init=None, voxel_size=1.0, max_range=100.0, min_range=0.5, max_iterations=50, threads=0, voxel_order_seed=0,
             exact_sums=False):
    """kiss_icp.rs:183-300 (KissIcpConfig defaults :40-49); correspondences index the voxel-downsampled source.
    Returns (IcpResult, number of downsampled source points)."""
    s, t = _f32(src, 3), _f32(tgt, 3)
    r = _IcpResult()
    cs = np.zeros(max(1, len(s)), np.uint64)
    ct = np.zeros(max(1, len(s)), np.uint64)
    r.corr_src = _p(cs, C.c_uint64)
    r.corr_tgt = _p(ct, C.c_uint64)
    i7 = _f32(IDENTITY if init is None else init).reshape(7)
    nd = C.c_size_t(0)
    lib().tco_set_voxel_order_seed(int(voxel_order_seed))      # the reference's HashMap order is unspecified: see tc_oracle.c
    lib().tco_set_exact_sums(1 if exact_sums else 0)            # diagnostic: the svd_transform sums' f32 terms added in f64
    try:
        rc = lib().tco_kiss_icp(_p(s), len(s), _p(t), len(t), _p(i7), voxel_size, max_range, min_range, max_iterations,
                                C.byref(r), C.byref(nd), threads)
    finally:
        lib().tco_set_voxel_order_seed(0)
        lib().tco_set_exact_sums(0)
    if rc:
        raise OracleError(rc)
    return IcpResult(r, len(s), cs, ct), int(nd.value)


def kiss_adaptive_threshold(init, voxel_size):
    return float(lib().tco_kiss_adaptive_threshold(_p(_f32(init).reshape(7)), voxel_size))


def voxel_grid_filter(points, voxel_size):
    """filtering.rs:38-133; output sorted by voxel key"""
    pts = _f32(points, 3)
    out = np.zeros((max(1, len(pts)), 3), np.float32)
    n_out = C.c_size_t(0)
    rc = lib().tco_voxel_grid_filter(_p(pts), len(pts), voxel_size, _p(out), C.byref(n_out))
    if rc:
        raise OracleError(rc)
    return out[: n_out.value].copy()


def p2plane_partial(src, j0, j1, tree, tgt_normals, T, max_dist=None):
    s, n = _f32(src, 3), _f32(tgt_normals, 3)
    out = np.zeros(29, np.float64)
    corr = np.zeros(max(1, j1 - j0), np.uint32)
    T7 = _f32(T).reshape(7)
    lib().tco_p2plane_partial(_p(s), j0, j1, tree._h, _p(tree.points), _p(n), _p(T7),
                              -1.0 if max_dist is None else max_dist, _p(out, C.c_double), _p(corr, C.c_uint32))
    return out, corr[: j1 - j0]


def p2p_partial(src, j0, j1, tree, T, max_dist=None):
    s = _f32(src, 3)
    out = np.zeros(17, np.float64)
    corr = np.zeros(max(1, j1 - j0), np.uint32)
    T7 = _f32(T).reshape(7)
    lib().tco_p2p_partial(_p(s), j0, j1, tree._h, _p(tree.points), _p(T7),
                          -1.0 if max_dist is None else max_dist, _p(out, C.c_double), _p(corr, C.c_uint32))
    return out, corr[: j1 - j0]


def symmetric_eigen3(m):
    a = _f32(m).reshape(9)
    ev, q = np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().tco_symmetric_eigen3(_p(a), _p(ev), _p(q))
    return ev, q.reshape(3, 3)


def symmetric_eigen3_batch(m):
    """(n, 3, 3) symmetric f32 -> (evals (n, 3), evecs (n, 3, 3) with eigenvectors as columns)"""
    a = _f32(m).reshape(-1, 9)
    ev, q = np.zeros((len(a), 3), np.float32), np.zeros((len(a), 9), np.float32)
    lib().tco_symmetric_eigen3_batch(_p(a), len(a), _p(ev), _p(q))
    return ev, q.reshape(-1, 3, 3)


def svd3(m):
    a = _f32(m).reshape(9)
    u, s, vt = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().tco_svd3(_p(a), _p(u), _p(s), _p(vt))
    return u.reshape(3, 3), s, vt.reshape(3, 3)


def cholesky6_solve(a, b):
    A, B, x = _f32(a).reshape(36), _f32(b).reshape(6), np.zeros(6, np.float32)
    ok = lib().tco_cholesky6_solve(_p(A), _p(B), _p(x))
    return (x if ok else None)


def lu6_solve(a, b):
    A, B, x = _f32(a).reshape(36), _f32(b).reshape(6), np.zeros(6, np.float32)
    ok = lib().tco_lu6_solve(_p(A), _p(B), _p(x))
    return (x if ok else None)


def quat_from_matrix(r):
    R, q = _f32(r).reshape(9), np.zeros(4, np.float32)
    lib().tco_quat_from_matrix(_p(R), _p(q))
    return q


def kabsch(s, q):
    S, Q = _f32(s, 3), _f32(q, 3)
    out = np.zeros(7, np.float32)
    ok = C.c_int(0)
    lib().tco_kabsch(_p(S), _p(Q), len(S), _p(out), C.byref(ok))
    return out


def isometry_apply(T, pts):
    T7 = _f32(T).reshape(7)
    P = _f32(pts, 3)
    out = np.zeros_like(P)
    L = lib()
    for i in range(len(P)):
        L.tco_isometry_apply(_p(T7), _p(P[i]), _p(out[i]))
    return out


def isometry_mul(a, b):
    A, B, o = _f32(a).reshape(7), _f32(b).reshape(7), np.zeros(7, np.float32)
    lib().tco_isometry_mul(_p(A), _p(B), _p(o))
    return o


def isometry_to_matrix(T):
    T7, m = _f32(T).reshape(7), np.zeros(16, np.float32)
    lib().tco_isometry_to_matrix(_p(T7), _p(m))
    return m.reshape(4, 4)


def multiscale_icp_point_to_point(src, tgt, init, levels, final_refinement_iterations, final_max_correspondence_distance,
                                  convergence_threshold, threads=0):
    """registration.rs:704-789 composed from the restated pieces (voxel_grid_filter + icp_point_to_point).
    levels = [(voxel_size, max_iterations, max_correspondence_distance or None), ...]"""
    s, t = _f32(src, 3), _f32(tgt, 3)
    if len(s) == 0 or len(t) == 0 or len(levels) == 0 or not (convergence_threshold > 0) or final_refinement_iterations == 0:
        raise OracleError(INVALID_DATA)
    cur = _f32(IDENTITY if init is None else init).reshape(7)
    total, last = 0, None
    for voxel, iters, md in levels:
        if not (voxel > 0) or iters == 0:
            raise OracleError(INVALID_DATA)
        sd, td = voxel_grid_filter(s, voxel), voxel_grid_filter(t, voxel)
        if len(sd) < 3 or len(td) < 3:
            continue
        last = icp_point_to_point(sd, td, cur, iters, convergence_threshold, md, threads)
        cur = last.transformation
        total += last.iterations
    if last is None:
        raise OracleError(ALGORITHM)
    fin = icp_point_to_point(s, t, cur, final_refinement_iterations, convergence_threshold, final_max_correspondence_distance, threads)
    fin.iterations += total
    return fin
