"""Uniform adapters so the restated reference tests run unchanged against the CPU oracle and the
HIP path (through the C ABI)."""
import numpy as np


class BackendError(Exception):
    def __init__(self, kind, msg=""):
        self.kind = kind
        super().__init__(f"{kind}: {msg}")


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def _wrap(self, fn, *a, **k):
        try:
            return fn(*a, **k)
        except self.O.OracleError as e:
            raise BackendError({1: "InvalidData", 2: "Algorithm"}[e.code], str(e))

    def estimate_normals(self, pts, k):
        return self._wrap(self.O.estimate_normals, pts, k)

    def estimate_normals_with_config(self, pts, k, radius=None, consistent_orientation=True, viewpoint=None):
        return self._wrap(self.O.estimate_normals, pts, k, radius, consistent_orientation, viewpoint)

    def estimate_normals_radius(self, pts, radius, consistent_orientation):
        return self._wrap(self.O.estimate_normals_radius, pts, radius, consistent_orientation)

    def icp_detailed(self, s, t, init, max_iters, max_dist, thr):
        return self._wrap(self.O.icp_detailed, s, t, init, max_iters, max_dist, thr)

    def icp_point_to_point(self, s, t, init, max_iters, thr, max_dist):
        return self._wrap(self.O.icp_point_to_point, s, t, init, max_iters, thr, max_dist)

    def icp(self, s, t, init, max_iters):
        return self.O.icp(s, t, init, max_iters)

    def icp_point_to_plane(self, s, t, n, init, max_iters):
        return self._wrap(self.O.icp_point_to_plane, s, t, n, init, max_iters)

    def icp_point_to_plane_detailed(self, s, t, n, init, max_iters, max_dist, thr):
        return self._wrap(self.O.icp_point_to_plane_detailed, s, t, n, init, max_iters, max_dist, thr)


    def gicp(self, s, t, init, max_iters=50, max_dist=1.0, thr=1e-6, k=20):
        return self._wrap(self.O.gicp, s, t, init, max_iters, max_dist, thr, k)

    def kiss_icp(self, s, t, init, voxel_size=1.0, max_range=100.0, min_range=0.5, max_iters=50):
        return self._wrap(lambda: self.O.kiss_icp(s, t, init, voxel_size, max_range, min_range, max_iters)[0])


class GpuBackend:
    name = "hip"

    def __init__(self, ctx):
        import threecrate_amd as tc
        self.tc, self.ctx = tc, ctx

    def _wrap(self, fn, *a, **k):
        tc = self.tc
        try:
            return fn(*a, **k)
        except tc.InvalidData as e:
            raise BackendError("InvalidData", str(e))
        except tc.AlgorithmError as e:
            raise BackendError("Algorithm", str(e))

    def estimate_normals(self, pts, k):
        return self._wrap(self.ctx.estimate_normals, pts, k)

    def estimate_normals_with_config(self, pts, k, radius=None, consistent_orientation=True, viewpoint=None):
        cfg = self.tc.NormalEstimationConfig(k, radius, consistent_orientation, viewpoint)
        return self._wrap(self.ctx.estimate_normals_with_config, pts, cfg)

    def estimate_normals_radius(self, pts, radius, consistent_orientation):
        return self._wrap(self.ctx.estimate_normals_radius, pts, radius, consistent_orientation)

    def icp_detailed(self, s, t, init, max_iters, max_dist, thr):
        return self._wrap(self.ctx.icp_detailed, s, t, init, max_iters, max_dist, thr)

    def icp_point_to_point(self, s, t, init, max_iters, thr, max_dist):
        return self._wrap(self.ctx.icp_point_to_point, s, t, init, max_iters, thr, max_dist)

    def icp(self, s, t, init, max_iters):
        return self.ctx.icp(s, t, init, max_iters)

    def icp_point_to_plane(self, s, t, n, init, max_iters):
        return self._wrap(self.ctx.icp_point_to_plane, s, t, n, init, max_iters)

    def icp_point_to_plane_detailed(self, s, t, n, init, max_iters, max_dist, thr):
        return self._wrap(self.ctx.icp_point_to_plane_detailed, s, t, n, init, max_iters, max_dist, thr)

    def gicp(self, s, t, init, max_iters=50, max_dist=1.0, thr=1e-6, k=20):
        import threecrate_amd as tc
        return self._wrap(self.ctx.gicp, s, t, init, tc.GicpConfig(max_iters, max_dist, thr, k))

    def kiss_icp(self, s, t, init, voxel_size=1.0, max_range=100.0, min_range=0.5, max_iters=50):
        import threecrate_amd as tc
        return self._wrap(self.ctx.kiss_icp, s, t, init, tc.KissIcpConfig(voxel_size, max_range, min_range, max_iters))


def empty_cloud():
    return np.zeros((0, 3), np.float32)
