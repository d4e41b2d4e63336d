"""`import threecrate_amd.compat as threecrate` -- the call shapes of the reference's Python module
(threecrate-python/src/lib.rs, stubs threecrate-python/threecrate.pyi) for the functions of this backend's path,
served by the HIP library.  Same class and function names, argument names, defaults, return shapes and exception
types (`RuntimeError` for algorithm / data errors like `to_py_err` lib.rs:40-42, `ValueError` for malformed arrays
lib.rs:85-128,~150-200, `IndexError` from `PointCloud.__getitem__`).

Covered: PointCloud, NormalPointCloud, IcpResult, KdTree, voxel_downsample, estimate_normals, icp,
icp_point_to_plane, gicp, kiss_icp, concatenate, transform_point_cloud.  Everything else of that module (meshes,
reconstruction, I/O formats, global registration, NDT, ROS messages) is outside SURVEY.md section 8.
"""
import numpy as np

from . import api as _api

__all__ = ["PointCloud", "NormalPointCloud", "IcpResult", "KdTree", "voxel_downsample", "estimate_normals", "icp",
           "icp_point_to_plane", "gicp", "kiss_icp", "concatenate", "transform_point_cloud"]


def _nx3(arr, what="Array"):
    """read_nx3_points (lib.rs): (N, 3) float32 or float64 -> float32"""
    if not isinstance(arr, np.ndarray) or arr.dtype not in (np.float32, np.float64):
        raise ValueError("Expected a numpy array of shape (N, 3) with dtype float32 or float64")
    if arr.ndim != 2 or arr.shape[1] != 3:
        raise ValueError(f"{what} must have shape (N, 3), got shape {list(arr.shape)} with dtype {arr.dtype}")
    return np.ascontiguousarray(arr, np.float32)


def _run(fn, *a, **k):
    try:
        return fn(*a, **k)
    except _api.Error as e:            # InvalidData / AlgorithmError / GpuError / Unsupported
        raise RuntimeError(str(e)) from None


def _isometry(mat):
    """numpy_to_isometry (lib.rs:76-137): 4 x 4 float32 / float64 -> (qi qj qk qw tx ty tz); None -> identity.
    The rotation is the closest rotation to the upper-left 3 x 3 block (what UnitQuaternion::from_matrix iterates
    to), here by polar decomposition in f64."""
    if mat is None:
        return None
    if not isinstance(mat, np.ndarray) or mat.dtype not in (np.float32, np.float64):
        raise ValueError("init_transform must be a 4×4 numpy array (float32 or float64)")
    if mat.shape != (4, 4):
        raise ValueError("init_transform must be a 4×4 array")
    m = mat.astype(np.float32).astype(np.float64)
    u, _, vt = np.linalg.svd(m[:3, :3])
    r = u @ np.diag([1.0, 1.0, np.sign(np.linalg.det(u @ vt)) or 1.0]) @ vt
    tr = r[0, 0] + r[1, 1] + r[2, 2]
    if tr > 0.0:
        d = np.sqrt(tr + 1.0) * 2.0
        w, i, j, k = 0.25 * d, (r[2, 1] - r[1, 2]) / d, (r[0, 2] - r[2, 0]) / d, (r[1, 0] - r[0, 1]) / d
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        d = np.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2.0
        w, i, j, k = (r[2, 1] - r[1, 2]) / d, 0.25 * d, (r[0, 1] + r[1, 0]) / d, (r[0, 2] + r[2, 0]) / d
    elif r[1, 1] > r[2, 2]:
        d = np.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2.0
        w, i, j, k = (r[0, 2] - r[2, 0]) / d, (r[0, 1] + r[1, 0]) / d, 0.25 * d, (r[1, 2] + r[2, 1]) / d
    else:
        d = np.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2.0
        w, i, j, k = (r[1, 0] - r[0, 1]) / d, (r[0, 2] + r[2, 0]) / d, (r[1, 2] + r[2, 1]) / d, 0.25 * d
    q = np.array([i, j, k, w]) / np.sqrt(i * i + j * j + k * k + w * w)
    return np.concatenate([q, m[:3, 3]]).astype(np.float32)


class PointCloud:
    """A 3D point cloud holding XYZ positions (lib.rs PyPointCloud)."""

    def __init__(self, arr=None):
        self._p = np.zeros((0, 3), np.float32) if arr is None else _nx3(arr)

    @staticmethod
    def from_numpy(arr):
        return PointCloud(arr)

    def to_numpy(self):
        return self._p.copy()

    @property
    def points(self):
        return self.to_numpy()

    @property
    def is_empty(self):
        return len(self._p) == 0

    def __len__(self):
        return len(self._p)

    def __getitem__(self, idx):
        n = len(self._p)
        i = n + idx if idx < 0 else idx
        if i < 0 or i >= n:
            raise IndexError("point cloud index out of range")
        return self._p[i].copy()

    def __add__(self, other):
        return PointCloud(np.concatenate([self._p, other._p]))

    def __array__(self, dtype=None, copy=None):
        a = self.to_numpy()
        return a if dtype is None else a.astype(dtype)

    def __repr__(self):
        return f"PointCloud({len(self._p)} points)"


class NormalPointCloud:
    """Positions + normals; returned by estimate_normals, accepted by icp_point_to_plane (lib.rs PyNormalPointCloud)."""

    def __init__(self, _pn=None):
        self._pn = np.zeros((0, 6), np.float32) if _pn is None else _pn

    @staticmethod
    def from_numpy(positions, normals):
        p, n = _nx3(positions, "Positions array"), _nx3(normals, "Normals array")
        if len(p) != len(n):
            raise ValueError(f"positions and normals must have the same length, got {len(p)} and {len(n)}")
        return NormalPointCloud(np.ascontiguousarray(np.concatenate([p, n], axis=1)))

    @property
    def is_empty(self):
        return len(self._pn) == 0

    def positions(self):
        return self._pn[:, 0:3].copy()

    def normals(self):
        return self._pn[:, 3:6].copy()

    def __len__(self):
        return len(self._pn)

    def __repr__(self):
        return f"NormalPointCloud({len(self._pn)} points)"


class IcpResult:
    """lib.rs:519-547"""

    def __init__(self, r):
        self._m = r.matrix
        self.mse = float(r.mse)
        self.iterations = int(r.iterations)
        self.converged = bool(r.converged)

    def transformation(self):
        """4 x 4 float32 rigid transform (source -> target)"""
        return self._m.copy()

    def __repr__(self):
        return f"IcpResult(converged={'true' if self.converged else 'false'}, mse={self.mse:.6f}, iterations={self.iterations})"


def _query3(query):
    if not isinstance(query, np.ndarray) or query.dtype not in (np.float32, np.float64) or query.ndim != 1:
        raise ValueError("Query point must be a 1D numpy array (float32 or float64)")
    if query.shape[0] != 3:
        raise ValueError("Query point must be a 1D array of length 3")
    return query.astype(np.float32)


class KdTree:
    """Spatial index over a cloud (lib.rs:707-776): built once (a device-resident grid index, tc_search_index_*), then
    queried; results as KdTree::find_k_nearest / find_radius_neighbors (nearest_neighbor.rs:177-298), radius results
    nearest first."""

    def __init__(self, cloud):
        self._n = len(cloud)                 # an empty cloud gives an empty tree (nearest_neighbor.rs:38-45)
        self._ix = _run(_api.SearchIndex, _api.default_context(), cloud._p)

    def knn(self, query, k):
        q = _query3(query)
        if k == 0 or self._n == 0:           # nearest_neighbor.rs:178-180
            return [], []
        pairs = _run(self._ix.find_k_nearest, q, min(int(k), self._n))
        return [i for i, _ in pairs], [d for _, d in pairs]

    def radius_search(self, query, radius):
        q = _query3(query)
        if self._n == 0:
            return [], []
        _, idx, dist = _run(self._ix.find_radius_neighbors_all, q.reshape(1, 3), float(radius))
        return [int(i) for i in idx], [float(d) for d in dist]

    def __repr__(self):
        return "KdTree"


def voxel_downsample(cloud, voxel_size):
    """voxel_grid_filter (filtering.rs:38-133); voxels come out sorted by (kx, ky, kz)"""
    return PointCloud(np.asarray(_run(_api.default_context().voxel_grid_filter, cloud._p, float(voxel_size)), np.float32))


def estimate_normals(cloud, k_neighbors=10):
    """lib.rs:827-833 -> estimate_normals (normals.rs:238-241)"""
    return NormalPointCloud(np.ascontiguousarray(_run(_api.default_context().estimate_normals, cloud._p, int(k_neighbors)), np.float32))


def icp(source, target, max_iterations=50, init_transform=None):
    """lib.rs:850-862 -> icp_point_to_point_default (registration.rs:683-701: threshold 1e-6, no distance limit)"""
    init = _isometry(init_transform)
    return IcpResult(_run(_api.default_context().icp_point_to_point, source._p, target._p, init, int(max_iterations), 1e-6, None,
                          correspondences=False))


def icp_point_to_plane(source, target, max_iterations=50, init_transform=None):
    """lib.rs:963-1007 -> icp_point_to_plane (registration.rs:488-494)"""
    init = _isometry(init_transform)
    return IcpResult(_run(_api.default_context().icp_point_to_plane, source._p, np.ascontiguousarray(target._pn[:, 0:3]),
                          np.ascontiguousarray(target._pn[:, 3:6]), init, int(max_iterations)))


def gicp(source, target, max_iterations=50, max_correspondence_distance=1.0, convergence_threshold=1e-6, k_correspondences=20,
         init_transform=None):
    """lib.rs:878-906 -> gicp (gicp.rs:100-305)"""
    init = _isometry(init_transform)
    cfg = _api.GicpConfig(int(max_iterations), float(max_correspondence_distance), float(convergence_threshold), int(k_correspondences))
    return IcpResult(_run(_api.default_context().gicp, source._p, target._p, init, cfg))


def kiss_icp(source, target, voxel_size=1.0, max_range=100.0, min_range=0.5, max_iterations=50, init_transform=None):
    """lib.rs:922-950 -> kiss_icp (kiss_icp.rs:183-300)"""
    init = _isometry(init_transform)
    cfg = _api.KissIcpConfig(float(voxel_size), float(max_range), float(min_range), int(max_iterations))
    return IcpResult(_run(_api.default_context().kiss_icp, source._p, target._p, init, cfg))


def concatenate(clouds):
    """lib.rs:1633-1642"""
    return PointCloud(np.concatenate([c._p for c in clouds]) if len(clouds) else np.zeros((0, 3), np.float32))


def transform_point_cloud(cloud, transform):
    """lib.rs:1660-1675: every point through the Isometry3 built from the 4 x 4 (quaternion form, f32)"""
    iso = _isometry(transform)
    if iso is None:
        raise ValueError("init_transform must be a 4×4 numpy array (float32 or float64)")
    q, t = iso[:4].astype(np.float32), iso[4:7].astype(np.float32)
    p = cloud._p
    qv = q[:3]
    t2 = np.cross(np.broadcast_to(qv, p.shape), p).astype(np.float32) * np.float32(2.0)
    out = ((t2 * q[3] + np.cross(np.broadcast_to(qv, p.shape), t2).astype(np.float32)) + p) + t
    return PointCloud(out.astype(np.float32))
