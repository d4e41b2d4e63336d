"""Synthetic workloads of SURVEY.md section 8(d) (counter-based SplitMix64, identical for the
CPU oracle and the HIP path).  Pure numpy; no device code.

The transforms mirror the reference bench driver: target = T * source with
T = (0.05, -0.02, 0.01) + 0.02 rad yaw (examples/threecrate_dataset_bench.rs:281-287).
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix_u01(seed, counters):
    """f32 in [0,1): z = seed + GOLDEN*(i+1); SplitMix64 finaliser; (z >> 40) * 2^-24."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GOLDEN * (counters.astype(np.uint64) + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def uniform_cloud(n, seed=1, scale=(1.0, 1.0, 1.0)):
    """N points uniform in [0,sx) x [0,sy) x [0,sz); coordinate c of point i uses counter 3i+c."""
    c = np.arange(3 * n, dtype=np.uint64)
    u = splitmix_u01(seed, c).reshape(n, 3)
    return (u * np.asarray(scale, dtype=np.float32)).astype(np.float32)


def yaw_isometry(t, yaw):
    """7-float isometry (qx qy qz qw tx ty tz): rotation about z by `yaw`, translation t."""
    h = np.float32(yaw) / np.float32(2.0)
    return np.array([0.0, 0.0, np.sin(h), np.cos(h), t[0], t[1], t[2]], dtype=np.float32)


def isometry_matrix(T):
    """4x4 float64 homogeneous matrix of a 7-float isometry."""
    x, y, z, w = [float(v) for v in T[:4]]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = [float(v) for v in T[4:7]]
    return M


def apply_isometry(T, pts):
    """float64 application, rounded to f32 once (generates inputs, not a parity-critical op)."""
    M = isometry_matrix(T)
    return (pts.astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32)


def invert_isometry(T):
    M = np.linalg.inv(isometry_matrix(T))
    return M


def harness_transform():
    """examples/threecrate_dataset_bench.rs:281-287"""
    return yaw_isometry((0.05, -0.02, 0.01), 0.02)


def small_transform(n):
    """T_small of SURVEY 8(d): t = s*(0.3,-0.2,0.1), yaw 0.1*s rad, s = n^(-1/3)."""
    s = float(n) ** (-1.0 / 3.0)
    return yaw_isometry((0.3 * s, -0.2 * s, 0.1 * s), 0.1 * s)


def registration_pair(n, seed=1, transform=None, scale=(1.0, 1.0, 1.0)):
    """target = cloud(seed); source = T^-1 * cloud, so that ICP(source -> target) recovers T."""
    tgt = uniform_cloud(n, seed, scale)
    T = small_transform(n) if transform is None else transform
    Minv = invert_isometry(T)
    src = (tgt.astype(np.float64) @ Minv[:3, :3].T + Minv[:3, 3]).astype(np.float32)
    return src, tgt, T
