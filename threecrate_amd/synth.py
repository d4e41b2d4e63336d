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


def gaussian_noise(n, seed, sigma):
    """(n, 3) f32 normal deviates, counter based like the clouds (Box-Muller over splitmix_u01): same bits everywhere."""
    c = np.arange(6 * n, dtype=np.uint64)
    u = splitmix_u01(seed, c).astype(np.float64).reshape(n, 3, 2)
    r = np.sqrt(-2.0 * np.log(np.maximum(u[..., 0], 2.0 ** -25)))
    return (sigma * r * np.cos(2.0 * np.pi * u[..., 1])).astype(np.float32)


def registration_pair(n, seed=1, transform=None, scale=(1.0, 1.0, 1.0), noise_sigma=0.0):
    """target = cloud(seed); source = T^-1 * cloud, so that ICP(source -> target) recovers T.
    noise_sigma > 0: INDEPENDENT sensor noise on both clouds (two scans of one scene never hold the same points): the
    converged phase of a registration then has non-zero nearest-neighbour distances, like real scan pairs."""
    tgt = uniform_cloud(n, seed, scale)
    T = small_transform(n) if transform is None else transform
    Minv = invert_isometry(T)
    src = (tgt.astype(np.float64) @ Minv[:3, :3].T + Minv[:3, 3]).astype(np.float32)
    if noise_sigma > 0.0:
        src = (src + gaussian_noise(n, 1000003 + 2 * seed, noise_sigma)).astype(np.float32)
        tgt = (tgt + gaussian_noise(n, 1000004 + 2 * seed, noise_sigma)).astype(np.float32)
    return src, tgt, T


# ---- scan-shaped synthetic clouds (BASELINE configs [2] and [4]) ----------
def _smooth_noise(u, v, seed):
    """cheap smooth 2-D field in [-1, 1] (sum of a few seeded sinusoids)."""
    rng = np.random.default_rng(seed)
    out = np.zeros_like(u, dtype=np.float64)
    for _ in range(6):
        fu, fv, ph = rng.uniform(0.5, 4.0), rng.uniform(0.5, 4.0), rng.uniform(0, 2 * np.pi)
        out += np.sin(fu * u + fv * v + ph)
    return out / 6.0


def tum_shaped_cloud(width=1155, height=866, seed=0, step=1):
    """Pinhole back-projection of a synthetic depth map z(u,v) = 1.5 + 0.5*smooth_noise (metres);
    fx = fy = 525*width/640, principal point centred (the reference's TUM loader uses fx=fy=525,
    cx=319.5, cy=239.5: examples/threecrate_dataset_bench.rs:339-357)."""
    us, vs = np.meshgrid(np.arange(0, width, step, dtype=np.float64), np.arange(0, height, step, dtype=np.float64))
    fx = 525.0 * width / 640.0
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    z = 1.5 + 0.5 * _smooth_noise(us / width * 6.0, vs / height * 6.0, seed)
    x = (us - cx) / fx * z
    y = (vs - cy) / fx * z
    return np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)


def kitti_shaped_cloud(beams=64, azimuth_steps=1875, seed=0, noise=0.02):
    """64 beams (elevation -24.8..+2 deg) x azimuth steps ray-cast onto the ground plane z = -1.73 m and
    four walls at +-20 m, range noise sigma = 2 cm."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(-24.8, 2.0, beams))[:, None]
    az = np.linspace(0, 2 * np.pi, azimuth_steps, endpoint=False)[None, :]
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el) * np.ones_like(az)], axis=-1).reshape(-1, 3)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_ground = np.where(d[:, 2] < -1e-6, -1.73 / d[:, 2], np.inf)
        t_wx = np.where(np.abs(d[:, 0]) > 1e-9, 20.0 / np.abs(d[:, 0]), np.inf)
        t_wy = np.where(np.abs(d[:, 1]) > 1e-9, 20.0 / np.abs(d[:, 1]), np.inf)
    t = np.minimum(np.minimum(t_ground, t_wx), t_wy)
    t = t + rng.normal(0.0, noise, t.shape)
    return (d * t[:, None]).astype(np.float32)
