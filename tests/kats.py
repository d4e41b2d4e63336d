"""Known-answer tests restated from the reference's own unit tests for the hot path
(SURVEY.md Appendix D).  Each function takes a backend (tests/backends.py) and asserts exactly
what the reference asserts; the docstring cites the reference test.

These are the only pins the reference holds for this path: tolerance-level property assertions,
no golden vectors.
"""
import numpy as np
import pytest

from tests.backends import BackendError, empty_cloud
from tests.helpers import quat_z, rotate_z, sphere_cloud

IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)


def _mag(v):
    return float(np.linalg.norm(np.asarray(v, np.float64)))


# ---- normals.rs:394-625 --------------------------------------------------------------------
def normals_simple(b):
    """normals.rs:399-422 test_estimate_normals_simple"""
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 0]], np.float32)
    r = b.estimate_normals(pts, 3)
    assert len(r) == 5
    assert np.all(np.abs(r[:, 5]) > 0.8)


def normals_empty(b):
    """normals.rs:425-429 test_estimate_normals_empty"""
    r = b.estimate_normals(empty_cloud(), 5)
    assert len(r) == 0


def normals_insufficient_k(b):
    """normals.rs:432-438 test_estimate_normals_insufficient_k"""
    with pytest.raises(BackendError) as e:
        b.estimate_normals(np.array([[0, 0, 0]], np.float32), 2)
    assert e.value.kind == "InvalidData"


def normals_empty_wins_over_bad_k(b):
    """normals.rs:261-269: the empty check precedes the k check"""
    assert len(b.estimate_normals(empty_cloud(), 2)) == 0


def normals_radius(b):
    """normals.rs:441-480 test_estimate_normals_radius"""
    pts = np.array([[np.float32(i) * np.float32(0.1), np.float32(j) * np.float32(0.1), 0.0]
                    for i in range(20) for j in range(20)], np.float32)
    r = b.estimate_normals_radius(pts, 0.2, True)
    assert len(r) == 400
    mags = np.linalg.norm(r[:, 3:], axis=1)
    assert np.all(np.abs(mags - 1.0) < 0.1)
    assert (np.abs(r[:, 5]) > 0.8).mean() * 100.0 > 80.0


def normals_cylinder(b):
    """normals.rs:483-548 test_estimate_normals_cylinder"""
    pts = []
    for i in range(10):
        for j in range(10):
            ang = np.float32(i) * np.float32(0.6)
            pts.append([np.cos(ang), np.sin(ang), np.float32(j) * np.float32(0.2) - np.float32(1.0)])
    pts = np.array(pts, np.float32)
    r = b.estimate_normals_with_config(pts, 8, None, True, (0.0, 0.0, 2.0))
    assert len(r) == 100
    mags = np.linalg.norm(r[:, 3:], axis=1)
    assert np.all(np.abs(mags - 1.0) < 0.1)
    assert (np.abs(r[:, 5]) < 0.8).mean() * 100.0 > 60.0


def normals_orientation_consistency(b):
    """normals.rs:551-592 test_estimate_normals_orientation_consistency"""
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    r = b.estimate_normals_with_config(pts, 3, None, True, (0.0, 0.0, 1.0))
    b.estimate_normals_with_config(pts, 3, None, False, None)
    first = r[0, 5]
    assert np.all(r[:, 5] * first > 0.0)


# ---- registration.rs:797-1140 (point-to-point) ------------------------------------------------
def icp_identity(b):
    """registration.rs:798-816 test_icp_identity_transformation"""
    pts = np.array([[i, 2 * i, 3 * i] for i in range(10)], np.float32)
    r = b.icp_detailed(pts, pts.copy(), IDENT, 10, None, 1e-6)
    assert r.converged and r.mse < 1e-6 and r.iterations <= 3


def icp_translation(b):
    """registration.rs:819-847 test_icp_translation"""
    s = np.array([[i, 2 * i, 3 * i] for i in range(10)], np.float32)
    t = s + np.array([1, 2, 3], np.float32)
    r = b.icp_detailed(s, t, IDENT, 50, None, 1e-6)
    assert _mag(r.transformation[4:]) > 0.05
    assert r.mse < 2.0


def icp_rotation(b):
    """registration.rs:850-870 test_icp_rotation"""
    s = np.array([[i, i % 5, 0] for i in range(20)], np.float32)
    t = rotate_z(s, np.pi / 4)
    r = b.icp_detailed(s, t, IDENT, 100, None, 1e-6)
    assert r.mse < 1.0


def icp_insufficient_points(b):
    """registration.rs:873-884 test_icp_insufficient_points"""
    with pytest.raises(BackendError) as e:
        b.icp_detailed(np.array([[0, 0, 0]], np.float32), np.array([[1, 1, 1]], np.float32), IDENT, 10, None, 1e-6)
    assert e.value.kind == "Algorithm"


def icp_api_compatibility(b):
    """registration.rs:887-903 test_icp_api_compatibility"""
    s = np.array([[i, i, 0] for i in range(5)], np.float32)
    t = s + np.array([1, 0, 0], np.float32)
    T = b.icp(s, t, IDENT, 20)
    assert _mag(T[4:]) > 0.5


def icp_swallows_errors(b):
    """registration.rs:238-241: any error returns `init`"""
    init = np.array([0, 0, 0.1, 0.99498743, 1, 2, 3], np.float32)
    T = b.icp(np.array([[0, 0, 0]], np.float32), np.array([[1, 1, 1]], np.float32), init, 10)
    assert np.array_equal(np.asarray(T, np.float32), init)
    T = b.icp(empty_cloud(), np.array([[1, 1, 1]], np.float32), init, 10)
    assert np.array_equal(np.asarray(T, np.float32), init)


def icp_p2p_basic(b):
    """registration.rs:928-957 test_icp_point_to_point_basic"""
    s = np.array([[x, y, z] for x in range(3) for y in range(3) for z in range(3)], np.float32)
    t = s + np.array([1.0, 0.5, 0.25], np.float32)
    r = b.icp_point_to_point(s, t, IDENT, 50, 1e-6, None)
    assert r.converged or r.iterations == 50
    assert r.mse < 2.0
    assert _mag(r.transformation[4:]) > 0.1


def icp_p2p_with_noise(b):
    """registration.rs:960-1002 test_icp_point_to_point_with_noise (noise drawn from a fixed seed)"""
    s = []
    for i in range(100):
        ang = np.float32(i) * np.float32(0.1)
        rad = np.float32(2.0) + np.float32(i % 10) * np.float32(0.1)
        s.append([rad * np.cos(ang), rad * np.sin(ang), np.float32(i % 5) * np.float32(0.5)])
    s = np.array(s, np.float32)
    t = rotate_z(s, 0.3) + np.array([2.0, 1.0, 0.5], np.float32)
    rng = np.random.default_rng(12345)
    t = (t + (rng.random((100, 3)).astype(np.float32) - np.float32(0.5)) * np.float32(0.1)).astype(np.float32)
    r = b.icp_point_to_point(s, t, IDENT, 100, 1e-5, None)
    assert r.mse < 0.5
    assert _mag(r.transformation[4:]) > 1.0


def icp_p2p_known_transform(b):
    """registration.rs:1005-1046 test_icp_point_to_point_known_transform"""
    s = np.array([[x, y, z] for x in range(-2, 3) for y in range(-2, 3) for z in range(-1, 2)], np.float32)
    known_t = np.array([1.0, -0.5, 0.25], np.float32)
    t = rotate_z(s, 0.2) + known_t
    r = b.icp_point_to_point(s, t, IDENT, 50, 1e-6, None)
    assert _mag(r.transformation[4:] - known_t) < 1.0
    assert r.mse < 0.5


def icp_p2p_convergence(b):
    """registration.rs:1049-1068 test_icp_point_to_point_convergence"""
    s = np.array([[np.float32(i) * np.float32(0.1), np.float32(i * 2) * np.float32(0.1), 0.0] for i in range(50)], np.float32)
    t = s + np.array([0.5, 0, 0], np.float32)
    r = b.icp_point_to_point(s, t, IDENT, 20, 1e-6, None)
    assert r.converged and r.iterations < 20 and r.mse < 0.1


def icp_p2p_max_distance(b):
    """registration.rs:1071-1098 test_icp_point_to_point_max_distance"""
    s = np.array([[i, 0, 0] for i in range(10)], np.float32)
    t = np.array([[i + 0.1, 0, 0] if i < 5 else [i + 10.0, 0, 0] for i in range(10)], np.float32)
    r = b.icp_point_to_point(s, t, IDENT, 20, 1e-6, 1.0)
    assert len(r.correspondences) <= 10
    assert r.mse < 5.0


def icp_p2p_default(b):
    """registration.rs:1101-1118 test_icp_point_to_point_default"""
    s = np.array([[i, i, 0] for i in range(10)], np.float32)
    t = s + np.array([1, 0, 0], np.float32)
    r = b.icp_point_to_point(s, t, IDENT, 30, 1e-6, None)
    assert r.mse < 1.0
    assert _mag(r.transformation[4:]) > 0.5


def icp_p2p_validation(b):
    """registration.rs:1121-1140 test_icp_point_to_point_validation"""
    t = np.array([[0, 0, 0]], np.float32)
    for args in [(empty_cloud(), t, IDENT, 10, 1e-6, None), (t, t, IDENT, 0, 1e-6, None), (t, t, IDENT, 10, -1e-6, None)]:
        with pytest.raises(BackendError) as e:
            b.icp_point_to_point(*args)
        assert e.value.kind == "InvalidData"
    with pytest.raises(BackendError) as e:      # icp_detailed: empty target / zero iterations (:266-276)
        b.icp_detailed(t, empty_cloud(), IDENT, 10, None, 1e-6)
    assert e.value.kind == "InvalidData"
    with pytest.raises(BackendError) as e:
        b.icp_detailed(t, t, IDENT, 0, None, 1e-6)
    assert e.value.kind == "InvalidData"


# ---- registration.rs:1142-1267 (point-to-plane) ---------------------------------------------
def p2pl_identity(b):
    """registration.rs:1168-1177 test_icp_point_to_plane_identity"""
    s, n = sphere_cloud(50)
    r = b.icp_point_to_plane(s, s.copy(), n, IDENT, 20)
    assert r.converged and r.mse < 1e-6


def p2pl_translation(b):
    """registration.rs:1180-1196 test_icp_point_to_plane_translation"""
    s, n = sphere_cloud(100)
    shift = np.array([0.15, 0, 0], np.float32)
    r = b.icp_point_to_plane(s, s + shift, n, IDENT, 50)
    assert _mag(r.transformation[4:] - shift) < 0.3
    assert r.mse < 0.1


def p2pl_validation(b):
    """registration.rs:1199-1216 test_icp_point_to_plane_validation"""
    s, n = sphere_cloud(20)
    with pytest.raises(BackendError) as e:
        b.icp_point_to_plane(s, s, np.array([[0, 0, 1]], np.float32), IDENT, 10)
    assert e.value.kind == "InvalidData"
    with pytest.raises(BackendError) as e:
        b.icp_point_to_plane(empty_cloud(), s, n, IDENT, 10)
    assert e.value.kind == "InvalidData"
    with pytest.raises(BackendError) as e:
        b.icp_point_to_plane_detailed(s, s, n, IDENT, 0, None, 1e-6)
    assert e.value.kind == "InvalidData"
    # precedence: normals-length mismatch is reported before max_iters == 0 (registration.rs:522-531)
    with pytest.raises(BackendError) as e:
        b.icp_point_to_plane_detailed(s, s, n[:3], IDENT, 0, None, 1e-6)
    assert e.value.kind == "InvalidData"


def p2pl_vs_p2pt(b):
    """registration.rs:1219-1251 test_icp_point_to_plane_vs_point_to_point_convergence"""
    s, n = sphere_cloud(80)
    t = s + np.array([0.1, 0.05, 0.0], np.float32)
    a = b.icp_point_to_plane(s, t, n, IDENT, 50)
    c = b.icp_point_to_point(s, t, IDENT, 50, 1e-6, None)
    assert _mag(a.transformation[4:]) > 0.05
    assert _mag(c.transformation[4:]) > 0.05
    assert a.converged or a.mse < 0.1


def p2pl_max_distance(b):
    """registration.rs:1254-1267 test_icp_point_to_plane_detailed_max_distance"""
    s, n = sphere_cloud(50)
    t = s + np.array([0.1, 0, 0], np.float32)
    r = b.icp_point_to_plane_detailed(s, t, n, IDENT, 30, 5.0, 1e-6)
    assert r.mse < 0.5


def p2pl_too_few_pairs(b):
    """registration.rs:568-572: fewer than 6 pairs -> Algorithm"""
    s, n = sphere_cloud(5)
    with pytest.raises(BackendError) as e:
        b.icp_point_to_plane(s, s.copy(), n, IDENT, 5)
    assert e.value.kind == "Algorithm"


# ---- gicp.rs:307-583 ----------------------------------------------------------------------------
def _sphere(n, radius):
    """make_sphere (gicp.rs:318-333 / kiss_icp.rs tests): Fibonacci sphere of the given radius"""
    pts, _ = sphere_cloud(n)
    return (pts / np.float32(3.0) * np.float32(radius)).astype(np.float32)


def _quat_axis(axis, angle):
    h = np.float32(angle) / np.float32(2)
    q = np.zeros(4, np.float32); q[axis] = np.sin(h); q[3] = np.cos(h)
    return q


def _rotate(q, pts):
    from oracle import oracle as O            # isometry apply of the checker side: test infrastructure
    T = np.concatenate([q, np.zeros(3, np.float32)]).astype(np.float32)
    M = O.isometry_to_matrix(T).astype(np.float64)
    return (pts.astype(np.float64) @ M[:3, :3].T).astype(np.float32)


def _angle_between(qa, qb):
    d = abs(float(np.dot(np.asarray(qa, np.float64) / np.linalg.norm(qa), np.asarray(qb, np.float64) / np.linalg.norm(qb))))
    return 2.0 * np.arccos(min(1.0, d))


def gicp_identity(b):
    """gicp.rs test gicp_identity_converges"""
    c = _sphere(100, 2.0)
    r = b.gicp(c, c, IDENT, 30)
    assert r.converged and r.mse < 1e-4


def gicp_small_translation(b):
    """gicp.rs test gicp_recovers_small_translation"""
    s = _sphere(150, 3.0)
    shift = np.array([0.1, 0.0, 0.0], np.float32)
    r = b.gicp(s, s + shift, IDENT, 60, 2.0)
    assert _mag(r.transformation[4:7] - shift) < 0.05 and r.mse < 0.1


def gicp_tiny_rotation(b):
    """gicp.rs test gicp_recovers_tiny_rotation_from_identity"""
    s = _sphere(300, 3.0)
    q = _quat_axis(2, np.deg2rad(2.0))
    r = b.gicp(s, _rotate(q, s), IDENT, 60, 0.8)
    assert _angle_between(r.transformation[:4], q) < np.deg2rad(0.5) and r.mse < 0.01


def gicp_rotation_from_init(b):
    """gicp.rs test gicp_refines_rotation_from_near_correct_init"""
    s = _sphere(200, 3.0)
    q = _quat_axis(2, np.deg2rad(8.0))
    init = np.concatenate([_quat_axis(2, np.deg2rad(6.0)), np.zeros(3, np.float32)]).astype(np.float32)
    r = b.gicp(s, _rotate(q, s), init, 60, 0.8)
    assert _angle_between(r.transformation[:4], q) < np.deg2rad(0.5)


def gicp_noise_and_outliers(b):
    """gicp.rs tests gicp_robust_to_gaussian_noise / gicp_robust_to_outlier_points"""
    s = _sphere(200, 3.0)
    i = np.arange(200, dtype=np.float32)
    noise = np.stack([np.sin(i * np.float32(1.6180339887)), np.cos(i * np.float32(2.7182818284)), np.sin(i * np.float32(3.1415926535))], 1) * np.float32(0.05)
    r = b.gicp(s, (s + noise).astype(np.float32), IDENT, 50, 1.0)
    assert r.mse < 0.05 and _mag(r.transformation[4:7]) < 0.1
    t = np.arange(20, dtype=np.float32)
    out = np.stack([t * np.float32(7.3) - 50, t * np.float32(3.1) - 30, t * np.float32(5.7) - 40], 1).astype(np.float32)
    r = b.gicp(s, np.concatenate([s, out]), IDENT, 40, 0.5)
    assert _mag(r.transformation[4:7]) < 0.05 and r.mse < 0.01


def gicp_validation(b):
    """gicp.rs tests gicp_empty_source_errors / zero_iterations / too_few_points / coplanar / result_fields_populated"""
    c = _sphere(30, 1.0)
    for args in ((empty_cloud(), c, IDENT), (c, c, IDENT, 0), (_sphere(10, 1.0), _sphere(10, 1.0), IDENT)):
        with pytest.raises(BackendError):
            b.gicp(*args)
    flat = np.array([[i * 0.1, j * 0.1, 0.0] for i in range(50) for j in range(50)], np.float32)
    with pytest.raises(BackendError):
        b.gicp(flat, flat, IDENT)
    c = _sphere(60, 2.0)
    r = b.gicp(c, c, IDENT, 10)
    assert r.iterations > 0 and len(r.correspondences) > 0


# ---- kiss_icp.rs:302-671 ------------------------------------------------------------------------
def _ring(n, rng_):
    a = np.float32(2.0) * np.float32(np.pi) * np.arange(n, dtype=np.float32) / np.float32(n)
    return np.stack([np.cos(a) * np.float32(rng_), np.sin(a) * np.float32(rng_), np.zeros(n, np.float32)], 1).astype(np.float32)


def _grid(n, spacing, z):
    side = int(np.ceil(np.sqrt(np.float32(n))))
    pts = [[i * spacing, j * spacing, z] for i in range(side) for j in range(side)][:n]
    return np.array(pts, np.float32)


def kiss_identity(b):
    """kiss_icp.rs test kiss_icp_identity_converges"""
    c = _ring(200, 5.0)
    r = b.kiss_icp(c, c, IDENT, 0.2, 50.0, 0.1, 30)
    assert r.converged or r.mse < 1e-4


def kiss_small_translation(b):
    """kiss_icp.rs test kiss_icp_recovers_small_translation"""
    s = _grid(100, 0.5, 5.0)
    shift = np.array([0.1, 0.0, 0.0], np.float32)
    r = b.kiss_icp(s, s + shift, IDENT, 0.2, 50.0, 0.1, 50)
    assert _mag(r.transformation[4:7] - shift) < 0.05 and r.mse < 0.1


def kiss_tiny_rotation(b):
    """kiss_icp.rs test kiss_icp_recovers_tiny_rotation_from_identity"""
    s = _sphere(300, 5.0)
    q = _quat_axis(2, np.deg2rad(3.0))
    r = b.kiss_icp(s, _rotate(q, s), IDENT, 0.5, 50.0, 0.1, 60)
    assert _angle_between(r.transformation[:4], q) < np.deg2rad(1.0)


def kiss_validation(b):
    """kiss_icp.rs tests kiss_icp_empty_source_errors / all_points_outside_range / zero_voxel_size / result_fields_populated"""
    c = _ring(30, 5.0)
    with pytest.raises(BackendError):
        b.kiss_icp(empty_cloud(), c, IDENT)
    with pytest.raises(BackendError):
        b.kiss_icp(_ring(50, 0.05), _ring(50, 0.05), IDENT, 1.0, 100.0, 0.5, 50)
    with pytest.raises(BackendError):
        b.kiss_icp(c, c, IDENT, 0.0)
    c = _ring(60, 5.0)
    r = b.kiss_icp(c, c, IDENT, 0.3, 50.0, 0.1, 10)
    assert r.iterations > 0 and len(r.correspondences) > 0


NORMALS_KATS = [normals_simple, normals_empty, normals_insufficient_k, normals_empty_wins_over_bad_k,
                normals_cylinder, normals_orientation_consistency]
NORMALS_RADIUS_KATS = [normals_radius]
ICP_KATS = [icp_identity, icp_translation, icp_rotation, icp_insufficient_points, icp_api_compatibility,
            icp_swallows_errors, icp_p2p_basic, icp_p2p_with_noise, icp_p2p_known_transform, icp_p2p_convergence,
            icp_p2p_max_distance, icp_p2p_default, icp_p2p_validation]
P2PL_KATS = [p2pl_identity, p2pl_translation, p2pl_validation, p2pl_vs_p2pt, p2pl_max_distance, p2pl_too_few_pairs]
GICP_KATS = [gicp_identity, gicp_small_translation, gicp_tiny_rotation, gicp_rotation_from_init, gicp_noise_and_outliers, gicp_validation]
KISS_KATS = [kiss_identity, kiss_small_translation, kiss_tiny_rotation, kiss_validation]
ALL_KATS = NORMALS_KATS + NORMALS_RADIUS_KATS + ICP_KATS + P2PL_KATS + GICP_KATS + KISS_KATS
