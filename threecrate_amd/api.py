"""Host-side mirror of the reference interface for the normals + ICP path.

Names, argument meaning and error behaviour follow threecrate-algorithms (normals.rs,
registration.rs) and the threecrate-gpu facade (device.rs, normals.rs, icp.rs); call shapes
follow the pyo3 module (threecrate-python/src/lib.rs:827-1010): N x 3 float32 in,
N x 6 / N x 3 normals and 4 x 4 float32 + mse + iterations + converged out.

Every function runs the HIP path through the C ABI (include/threecrate_hip.h).  numpy inputs
take the host entry points (H2D staging inside the library); torch CUDA tensors take the
*_device entry points with zero copies.  There is no CPU fallback.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _lib

IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float32)


# ---- threecrate_core::Error (threecrate-core/src/error.rs:7-28) ------------------------------
class Error(Exception):
    pass


class InvalidData(Error):
    pass


class AlgorithmError(Error):
    pass


class GpuError(Error):
    pass


class Unsupported(Error):
    pass


_ERR = {_lib.TC_INVALID_DATA: InvalidData, _lib.TC_ALGORITHM: AlgorithmError, _lib.TC_GPU: GpuError,
        _lib.TC_UNSUPPORTED: Unsupported}


@dataclass
class NormalEstimationConfig:
    """normals.rs:17-37"""
    k_neighbors: int = 10
    radius: Optional[float] = None
    consistent_orientation: bool = True
    viewpoint: Optional[tuple] = None


@dataclass
class IcpScaleLevel:
    """registration.rs:27-35"""
    voxel_size: float
    max_iterations: int
    max_correspondence_distance: Optional[float] = None


@dataclass
class MultiScaleIcpConfig:
    """registration.rs:38-71 (same defaults)"""
    levels: list = field(default_factory=lambda: [IcpScaleLevel(0.20, 10, 0.50), IcpScaleLevel(0.10, 10, 0.25),
                                                  IcpScaleLevel(0.05, 15, 0.15)])
    final_refinement_iterations: int = 10
    final_max_correspondence_distance: Optional[float] = 0.10
    convergence_threshold: float = 1e-5


@dataclass
class GicpConfig:
    """gicp.rs:25-40"""
    max_iterations: int = 50
    max_correspondence_distance: float = 1.0
    convergence_threshold: float = 1e-6
    k_correspondences: int = 20


@dataclass
class KissIcpConfig:
    """kiss_icp.rs:28-49"""
    voxel_size: float = 1.0
    max_range: float = 100.0
    min_range: float = 0.5
    max_iterations: int = 50


@dataclass
class ICPResult:
    """registration.rs:13-24; `transformation` is the 7-float Isometry3 (qi qj qk qw tx ty tz)."""
    transformation: np.ndarray
    mse: float
    iterations: int
    converged: bool
    correspondences: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int64))
    corr_target: object = None   # dense per-source target index (0xFFFFFFFF = none), numpy or torch

    @property
    def matrix(self):
        """4 x 4 float32 homogeneous matrix (threecrate-python/src/lib.rs:48-61)."""
        return isometry_to_matrix(self.transformation)


def isometry_to_matrix(T):
    x, y, z, w = [np.float32(v) for v in T[:4]]
    two = np.float32(2)
    ww, xx, yy, zz = w * w, x * x, y * y, z * z
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = ww + xx - yy - zz; m[0, 1] = x * y * two - w * z * two; m[0, 2] = w * y * two + x * z * two
    m[1, 0] = w * z * two + x * y * two; m[1, 1] = ww - xx + yy - zz; m[1, 2] = y * z * two - w * x * two
    m[2, 0] = x * z * two - w * y * two; m[2, 1] = w * x * two + y * z * two; m[2, 2] = ww - xx - yy + zz
    m[:3, 3] = T[4:7]
    return m


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _as_host(a, cols=3):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if a.size == 0:
        return a.reshape(0, cols)
    return a.reshape(-1, cols)


class GpuContext:
    """GpuContext::new (threecrate-gpu/src/device.rs:16-50): one HIP device + one stream."""

    def __init__(self, device: int = 0, stream=None):
        self._L = _lib.load()
        h = C.c_void_p()
        if stream is None:
            rc = self._L.tc_context_create(device, C.byref(h))
        else:
            rc = self._L.tc_context_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if rc != _lib.TC_OK:
            raise GpuError(f"no usable HIP device {device} (tc_status {rc}); threecrate_amd has no CPU fallback")
        self._h = h
        self.device = device
        self.stream = stream          # raw hipStream_t the context enqueues on (None: its own stream)

    def trim(self):
        """tc_context_trim: release the device memory parked by destroyed handles (the context keeps a few handles' worth for
        reuse) -- before handing the GPU to another allocator"""
        self._check(self._L.tc_context_trim(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.tc_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != _lib.TC_OK:
            msg = self._L.tc_last_error_message(self._h).decode()
            raise _ERR.get(rc, Error)(msg)

    def _order(self, device):
        """Stream ordering rule of the *_device entry points (include/threecrate_hip.h): the library reads the
        tensors on the context's stream, torch produced them (conversions, slices, the caller's own ops) on its
        current stream -> make the context's stream wait for torch's (an event, no host wait)."""
        import torch
        cur = torch.cuda.current_stream(device).cuda_stream
        if self.stream is None or self.stream != cur:
            self._check(self._L.tc_context_wait_stream(self._h, C.c_void_p(cur)))

    def _release(self, device):
        """the reverse of _order: torch's current stream waits for what the context's stream holds now (a buffer torch owns
        was handed to a stream-ordered entry point and may be overwritten or freed by torch next)"""
        import torch
        cur = torch.cuda.current_stream(device).cuda_stream
        if self.stream is None or self.stream != cur:
            self._check(self._L.tc_stream_wait_context(self._h, C.c_void_p(cur)))

    @staticmethod
    def _max_dist(d):
        """Option<f32> -> the ABI's encoding (< 0 = None).  A negative Some(d) must not alias None (see _reject_all):
        it is returned as None."""
        if d is None:
            return -1.0
        return None if float(d) < 0.0 else float(d)

    @staticmethod
    def _reject_all(ns, nt, max_iters, normals_len=None):
        """A negative Some(max_correspondence_distance): the reference rejects every pair (registration.rs:100-101,
        `distance > d` always holds), i.e. after its validation (:266-276 / :517-531) the first iteration fails with
        "Insufficient correspondences" (:311-315 / :568-572)."""
        if ns == 0 or nt == 0:
            raise InvalidData("Source or target point cloud is empty")
        if normals_len is not None and normals_len != nt:
            raise InvalidData("target_normals length must equal the number of target points")
        if max_iters == 0:
            raise InvalidData("Max iterations must be positive")
        raise AlgorithmError("Insufficient correspondences found (negative max_correspondence_distance rejects every pair)")

    # ---- profiling ----
    def profile_enable(self, on=True):
        """False/0 off, True/1 every kernel, 2 = sampled events on the dominant kernel only, 3 = no events: the ICP main pass
        counts its searches instead (search_stats())."""
        self._L.tc_profile_enable(self._h, int(on))

    def profile_reset(self):
        self._L.tc_profile_reset(self._h)

    def debug_counter(self, which="indexed_points"):
        """work counters since the context was created (tc_debug_counter): "indexed_points" = points that went through an index
        build, "index_builds" = the builds"""
        return int(self._L.tc_debug_counter(self._h, {"indexed_points": _lib.TC_COUNTER_INDEXED_POINTS, "index_builds": _lib.TC_COUNTER_INDEX_BUILDS}[which]))

    def search_stats(self):
        """Search statistics of the ICP main pass over the registrations run since profile_enable(3) (SURVEY 8d's secondary
        figures; tc_debug_counter TC_COUNTER_ICP_*): raw counters + the derived per-search / lock-step figures.  A candidate step
        = four consecutive target records = four distance evaluations of one lane."""
        c = {k: int(self._L.tc_debug_counter(self._h, v)) for k, v in (
            ("iterations", _lib.TC_COUNTER_ICP_ITERATIONS), ("wave_trips", _lib.TC_COUNTER_ICP_TRIPS),
            ("wave_trips_without_a_search", _lib.TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH), ("searches", _lib.TC_COUNTER_ICP_SEARCHES),
            ("candidate_steps_needed", _lib.TC_COUNTER_ICP_STEPS_NEEDED), ("candidate_steps_taken_by_slowest_lanes", _lib.TC_COUNTER_ICP_STEPS_TAKEN))}
        s, need, took = max(c["searches"], 1), max(c["candidate_steps_needed"], 1), c["candidate_steps_taken_by_slowest_lanes"]
        c["candidates_per_search"] = 4.0 * c["candidate_steps_needed"] / s
        c["distance_evaluations"] = 4 * c["candidate_steps_needed"]
        c["lockstep_ratio"] = 64.0 * took / need          # lane slots the trips spent per lane step needed
        c["steps_per_searching_trip"] = took / max(c["wave_trips"] - c["wave_trips_without_a_search"], 1)
        c["steps_per_search"] = c["candidate_steps_needed"] / s
        return c

    def profile_read(self, minmax=False):
        """{kernel name: (launches, total ms)}; minmax=True: (launches, total ms, shortest launch ms, longest launch ms)"""
        buf = (_lib.KernelStatC * 64)()
        n = self._L.tc_profile_read(self._h, buf, 64)
        if minmax:
            return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms), float(buf[i].min_ms), float(buf[i].max_ms)) for i in range(min(n, 64))}
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(n, 64))}

    # ---- normals ----
    def _cfg(self, config: NormalEstimationConfig):
        c = _lib.NormalConfig()
        self._L.tc_normal_config_default(C.byref(c))
        c.k_neighbors = int(config.k_neighbors)
        if config.radius is not None:
            c.has_radius, c.radius = 1, float(config.radius)
        c.consistent_orientation = 1 if config.consistent_orientation else 0
        if config.viewpoint is not None:
            c.has_viewpoint = 1
            for i in range(3):
                c.viewpoint[i] = float(config.viewpoint[i])
        return c

    def estimate_normals_with_config(self, cloud, config: NormalEstimationConfig):
        """normals.rs:257-357 -> (N, 6) array of NormalPoint3f {position, normal}."""
        c = self._cfg(config)
        if _is_torch(cloud):
            import torch
            x = cloud.detach().to(torch.float32).contiguous().reshape(-1, 3)
            out = torch.empty((x.shape[0], 6), dtype=torch.float32, device=x.device)
            self._order(x.device)
            self._check(self._L.tc_estimate_normals_device(self._h, x.data_ptr(), x.shape[0], C.byref(c), out.data_ptr()))
            return out
        x = _as_host(cloud)
        out = np.empty((x.shape[0], 6), np.float32)
        self._check(self._L.tc_estimate_normals(self._h, x.ctypes.data, x.shape[0], C.byref(c), out.ctypes.data))
        return out

    def estimate_normals_slice(self, cloud, config: NormalEstimationConfig, begin: int, end: int):
        """tc_estimate_normals_slice_device: NormalPoint3f records of the cell-sorted positions [begin, end) of the
        device-resident cloud, in sorted order (one rank's share of a multi-GPU run, threecrate_amd.distributed)."""
        import torch
        c = self._cfg(config)
        x = cloud.detach().to(torch.float32).contiguous().reshape(-1, 3)
        out = torch.empty((max(end - begin, 0), 6), dtype=torch.float32, device=x.device)
        self._order(x.device)
        self._check(self._L.tc_estimate_normals_slice_device(self._h, x.data_ptr(), x.shape[0], C.byref(c), int(begin), int(end), out.data_ptr()))
        return out

    def normals_unsort(self, sorted_all):
        """tc_normals_unsort_device: the gathered slices (n, 6, sorted order) -> (n, 6) in input order; uses the index the
        last estimate_normals_slice call left in this context."""
        import torch
        srt = sorted_all.detach().to(torch.float32).contiguous()
        out = torch.empty_like(srt)
        self._order(srt.device)
        self._check(self._L.tc_normals_unsort_device(self._h, srt.data_ptr(), srt.shape[0], out.data_ptr()))
        return out

    def estimate_normals(self, cloud, k: int = 10):
        """normals.rs:238-247"""
        return self.estimate_normals_with_config(cloud, NormalEstimationConfig(k_neighbors=k))

    def estimate_normals_radius(self, cloud, radius: float, consistent_orientation: bool):
        """normals.rs:368-380"""
        return self.estimate_normals_with_config(
            cloud, NormalEstimationConfig(k_neighbors=10, radius=radius, consistent_orientation=consistent_orientation))

    def compute_normals(self, points, k: int):
        """GpuContext::compute_normals (threecrate-gpu/src/normals.rs:367-374): normals only (N, 3)."""
        return self.estimate_normals(points, k)[:, 3:]

    # ---- batch k-NN ----
    def find_k_nearest_batch(self, cloud, queries, k: int):
        """gpu_find_k_nearest_batch (threecrate-gpu/src/nearest_neighbor.rs:345-355) /
        KdTree.knn (threecrate-python/src/lib.rs:735-745): (idx (nq,k) int64, dist (nq,k) f32, count (nq,)).
        Rows are ascending by distance; entries past count[q] are undefined."""
        c, q = _as_host(cloud), _as_host(queries)
        kk = max(int(k), 1)
        idx = np.zeros((len(q), kk), np.uint32)
        dist = np.zeros((len(q), kk), np.float32)
        cnt = np.zeros(len(q), np.uint32)
        self._check(self._L.tc_knn(self._h, c.ctypes.data, c.shape[0], q.ctypes.data, q.shape[0], int(k), idx.ctypes.data,
                                   dist.ctypes.data, cnt.ctypes.data))
        return idx.astype(np.int64), dist, cnt

    def find_k_nearest(self, cloud, query, k: int):
        """gpu_find_k_nearest (threecrate-gpu/src/nearest_neighbor.rs:332-343): [(index, distance), ...]"""
        idx, dist, cnt = self.find_k_nearest_batch(cloud, np.asarray(query, np.float32).reshape(1, 3), k)
        return [(int(idx[0, i]), float(dist[0, i])) for i in range(int(cnt[0]))]

    def find_radius_neighbors_batch(self, cloud, queries, radius: float, k_max: int = 32):
        """find_radius_neighbors (nearest_neighbor.rs:254-298) for many queries, capped at the k_max nearest like
        gpu_find_radius_neighbors (threecrate-gpu/src/nearest_neighbor.rs:357-367): (idx, dist, count)."""
        c, q = _as_host(cloud), _as_host(queries)
        kk = max(int(k_max), 1)
        idx = np.zeros((len(q), kk), np.uint32)
        dist = np.zeros((len(q), kk), np.float32)
        cnt = np.zeros(len(q), np.uint32)
        self._check(self._L.tc_radius_search(self._h, c.ctypes.data, c.shape[0], q.ctypes.data, q.shape[0], float(radius), int(k_max),
                                             idx.ctypes.data, dist.ctypes.data, cnt.ctypes.data))
        return idx.astype(np.int64), dist, cnt

    def find_radius_neighbors(self, cloud, query, radius: float, k_max: int = 32):
        """gpu_find_radius_neighbors (threecrate-gpu/src/nearest_neighbor.rs:357-367): [(index, distance), ...]"""
        idx, dist, cnt = self.find_radius_neighbors_batch(cloud, np.asarray(query, np.float32).reshape(1, 3), radius, k_max)
        return [(int(idx[0, i]), float(dist[0, i])) for i in range(int(cnt[0]))]

    # ---- voxel grid filter ----
    def voxel_grid_filter(self, cloud, voxel_size: float):
        """filtering.rs:38-133 -> (M, 3) centroids, sorted by voxel key (kx, ky, kz)."""
        n_out = C.c_size_t(0)
        if _is_torch(cloud):
            import torch
            x = cloud.detach().to(torch.float32).contiguous().reshape(-1, 3)
            out = torch.empty((max(1, x.shape[0]), 3), dtype=torch.float32, device=x.device)
            self._order(x.device)
            self._check(self._L.tc_voxel_grid_filter_device(self._h, x.data_ptr(), x.shape[0], voxel_size, out.data_ptr(), C.byref(n_out)))
            return out[: n_out.value]
        x = _as_host(cloud)
        out = np.empty((max(1, x.shape[0]), 3), np.float32)
        self._check(self._L.tc_voxel_grid_filter(self._h, x.ctypes.data, x.shape[0], voxel_size, out.ctypes.data, C.byref(n_out)))
        return out[: n_out.value].copy()

    # ---- ICP ----
    def _result(self, r, ns, corr, want_pairs):
        T = np.array(list(r.transformation), np.float32)
        res = ICPResult(T, float(r.mse), int(r.iterations), bool(r.converged), corr_target=corr)
        if want_pairs is True and corr is not None:
            ct = corr.cpu().numpy() if _is_torch(corr) else corr
            ct = ct.astype(np.int64)
            src = np.nonzero(ct != 0xFFFFFFFF)[0]
            res.correspondences = np.stack([src, ct[src]], axis=1)
        return res

    def icp_detailed(self, source, target, init=None, max_iters=50, max_correspondence_distance=None,
                     convergence_threshold=1e-6, correspondences=True, _checked=False):
        """registration.rs:258-370"""
        md = self._max_dist(max_correspondence_distance)
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        r = _lib.IcpResultC()
        if _is_torch(source):
            import torch
            s = source.detach().to(torch.float32).contiguous().reshape(-1, 3)
            t = target.detach().to(torch.float32).contiguous().reshape(-1, 3)
            corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
            r.corr_target = corr.data_ptr() if corr is not None else None
            if md is None:
                self._reject_all(s.shape[0], t.shape[0], max_iters)
            if _checked and not (convergence_threshold > 0):
                raise InvalidData("Convergence threshold must be positive")
            self._order(s.device)
            self._check(self._L.tc_icp_detailed_device(self._h, s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0],
                                                       i7.ctypes.data, max_iters, md, convergence_threshold, C.byref(r)))
            if corr is not None:
                # correspondences="device": the dense per-source target index stays as written (int32 bits, -1 = none)
                corr = corr[: s.shape[0]] if correspondences == "device" else corr[: s.shape[0]].to(torch.int64) & 0xFFFFFFFF
            return self._result(r, s.shape[0], corr, correspondences)
        s, t = _as_host(source), _as_host(target)
        corr = np.empty(max(1, s.shape[0]), np.uint32) if correspondences else None
        r.corr_target = corr.ctypes.data if corr is not None else None
        if md is None:
            self._reject_all(s.shape[0], t.shape[0], max_iters)
        fn = self._L.tc_icp_point_to_point if _checked else self._L.tc_icp_detailed
        a, b = (convergence_threshold, md) if _checked else (md, convergence_threshold)
        self._check(fn(self._h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], i7.ctypes.data, max_iters, a, b, C.byref(r)))
        return self._result(r, s.shape[0], None if corr is None else corr[: s.shape[0]], correspondences)

    def gicp(self, source, target, init=None, config: "GicpConfig" = None, correspondences=True):
        """gicp.rs:100-305"""
        cfg = config or GicpConfig()
        c = _lib.GicpConfigC(cfg.max_iterations, cfg.max_correspondence_distance, cfg.convergence_threshold, cfg.k_correspondences)
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        r = _lib.IcpResultC()
        if _is_torch(source):
            import torch
            s = source.detach().to(torch.float32).contiguous().reshape(-1, 3)
            t = target.detach().to(torch.float32).contiguous().reshape(-1, 3)
            corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
            r.corr_target = corr.data_ptr() if corr is not None else None
            self._order(s.device)
            self._check(self._L.tc_gicp_device(self._h, s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0], i7.ctypes.data, C.byref(c), C.byref(r)))
            if corr is not None:
                # correspondences="device": the dense per-source target index stays as written (int32 bits, -1 = none)
                corr = corr[: s.shape[0]] if correspondences == "device" else corr[: s.shape[0]].to(torch.int64) & 0xFFFFFFFF
            return self._result(r, s.shape[0], corr, correspondences)
        s, t = _as_host(source), _as_host(target)
        corr = np.empty(max(1, s.shape[0]), np.uint32) if correspondences else None
        r.corr_target = corr.ctypes.data if corr is not None else None
        self._check(self._L.tc_gicp(self._h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], i7.ctypes.data, C.byref(c), C.byref(r)))
        return self._result(r, s.shape[0], None if corr is None else corr[: s.shape[0]], correspondences)

    def kiss_icp(self, source, target, init=None, config: "KissIcpConfig" = None, correspondences=True):
        """kiss_icp.rs:183-300.  `correspondences` pairs (index into the voxel-downsampled source, target index)."""
        cfg = config or KissIcpConfig()
        c = _lib.KissIcpConfigC(cfg.voxel_size, cfg.max_range, cfg.min_range, cfg.max_iterations)
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        r = _lib.IcpResultC()
        nd = C.c_size_t(0)
        if _is_torch(source):
            import torch
            s = source.detach().to(torch.float32).contiguous().reshape(-1, 3)
            t = target.detach().to(torch.float32).contiguous().reshape(-1, 3)
            corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
            r.corr_target = corr.data_ptr() if corr is not None else None
            self._order(s.device)
            self._check(self._L.tc_kiss_icp_device(self._h, s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0], i7.ctypes.data,
                                                   C.byref(c), C.byref(r), C.byref(nd)))
            if corr is not None:
                corr = corr[: nd.value].to(torch.int64) & 0xFFFFFFFF
            return self._result(r, nd.value, corr, correspondences)
        s, t = _as_host(source), _as_host(target)
        corr = np.empty(max(1, s.shape[0]), np.uint32) if correspondences else None
        r.corr_target = corr.ctypes.data if corr is not None else None
        self._check(self._L.tc_kiss_icp(self._h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], i7.ctypes.data,
                                        C.byref(c), C.byref(r), C.byref(nd)))
        return self._result(r, nd.value, None if corr is None else corr[: nd.value], correspondences)

    def icp_point_to_point(self, source, target, init=None, max_iterations=50, convergence_threshold=1e-6,
                           max_correspondence_distance=None, correspondences=True):
        """registration.rs:644-680 (adds the threshold > 0 check)"""
        return self.icp_detailed(source, target, init, max_iterations, max_correspondence_distance,
                                 convergence_threshold, correspondences, _checked=True)

    def icp(self, source, target, init=None, max_iters=50):
        """registration.rs:232-242: returns the 7-float isometry; any error returns `init`."""
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        if _is_torch(source):
            try:
                return self.icp_detailed(source, target, i7, max_iters, None, 1e-6, correspondences=False).transformation
            except Error:
                return i7.copy()
        s, t = _as_host(source), _as_host(target)
        out = np.zeros(7, np.float32)
        self._check(self._L.tc_icp(self._h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], i7.ctypes.data, max_iters,
                                   out.ctypes.data))
        return out

    def multiscale_icp_point_to_point(self, source, target, init=None, config=None):
        """registration.rs:704-789"""
        cfg = config or MultiScaleIcpConfig()
        s, t = _as_host(source), _as_host(target)
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        lv = (_lib.ScaleLevelC * max(1, len(cfg.levels)))()
        for i, l in enumerate(cfg.levels):
            lv[i].voxel_size, lv[i].max_iterations = float(l.voxel_size), int(l.max_iterations)
            lv[i].max_correspondence_distance = -1.0 if l.max_correspondence_distance is None else float(l.max_correspondence_distance)
        c = _lib.MultiScaleConfigC(lv, len(cfg.levels), int(cfg.final_refinement_iterations),
                                   -1.0 if cfg.final_max_correspondence_distance is None else float(cfg.final_max_correspondence_distance),
                                   float(cfg.convergence_threshold))
        r = _lib.IcpResultC()
        corr = np.empty(max(1, s.shape[0]), np.uint32)
        r.corr_target = corr.ctypes.data
        self._check(self._L.tc_multiscale_icp_point_to_point(self._h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0],
                                                             i7.ctypes.data, C.byref(c), C.byref(r)))
        return self._result(r, s.shape[0], corr[: s.shape[0]], True)

    def icp_point_to_plane_detailed(self, source, target, target_normals, init=None, max_iters=50,
                                    max_correspondence_distance=None, convergence_threshold=1e-6, correspondences=True):
        """registration.rs:508-602.  target_normals: (Nt, 3) Vector3f, or the (Nt, 6) NormalPoint3f
        array returned by estimate_normals (its normal columns are used in place, stride 6)."""
        md = self._max_dist(max_correspondence_distance)
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        r = _lib.IcpResultC()
        if _is_torch(source):
            import torch
            s = source.detach().to(torch.float32).contiguous().reshape(-1, 3)
            t = target.detach().to(torch.float32).contiguous().reshape(-1, 3)
            n = target_normals.detach().to(torch.float32).contiguous()
            stride = 6 if (n.dim() == 2 and n.shape[1] == 6) else 3
            nptr = n.data_ptr() + (12 if stride == 6 else 0)
            nn = n.shape[0] if n.dim() == 2 else n.numel() // 3
            if md is None:
                self._reject_all(s.shape[0], t.shape[0], max_iters, nn)
            corr = torch.empty(max(1, s.shape[0]), dtype=torch.int32, device=s.device) if correspondences else None
            r.corr_target = corr.data_ptr() if corr is not None else None
            self._order(s.device)
            self._check(self._L.tc_icp_point_to_plane_detailed_device(
                self._h, s.data_ptr(), s.shape[0], t.data_ptr(), t.shape[0], nptr, nn, stride, i7.ctypes.data, max_iters,
                md, convergence_threshold, C.byref(r)))
            if corr is not None:
                # correspondences="device": the dense per-source target index stays as written (int32 bits, -1 = none)
                corr = corr[: s.shape[0]] if correspondences == "device" else corr[: s.shape[0]].to(torch.int64) & 0xFFFFFFFF
            return self._result(r, s.shape[0], corr, correspondences)
        s, t = _as_host(source), _as_host(target)
        n = np.ascontiguousarray(np.asarray(target_normals, np.float32))
        stride = 6 if (n.ndim == 2 and n.shape[1] == 6) else 3
        nn = n.shape[0] if n.ndim == 2 else n.size // 3
        nptr = n.ctypes.data + (12 if stride == 6 else 0)
        if md is None:
            self._reject_all(s.shape[0], t.shape[0], max_iters, nn)
        corr = np.empty(max(1, s.shape[0]), np.uint32) if correspondences else None
        r.corr_target = corr.ctypes.data if corr is not None else None
        self._check(self._L.tc_icp_point_to_plane_detailed(
            self._h, s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0], nptr, nn, stride, i7.ctypes.data, max_iters, md,
            convergence_threshold, C.byref(r)))
        return self._result(r, s.shape[0], None if corr is None else corr[: s.shape[0]], correspondences)

    def icp_point_to_plane(self, source, target, target_normals, init=None, max_iters=50):
        """registration.rs:488-496"""
        return self.icp_point_to_plane_detailed(source, target, target_normals, init, max_iters, None, 1e-6)


def _hip_memcpy_dtoh(dst, src, nbytes):
    """hipMemcpy(dst, src, nbytes, hipMemcpyDeviceToHost) through the HIP runtime the library is bound to (already in the process'
    global namespace: _lib._preload_hip_runtime / the library's own NEEDED entry) -- never a second copy of the runtime"""
    fn = None
    try:
        fn = C.CDLL(None).hipMemcpy
    except AttributeError:
        # the runtime was loaded without RTLD_GLOBAL (no torch preload: it came in as the library's own NEEDED entry): take the
        # copy that is ALREADY in the process by its soname -- RTLD_NOLOAD never loads a second one
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                fn = C.CDLL(name, mode=os.RTLD_NOLOAD | os.RTLD_NOW).hipMemcpy
                break
            except (OSError, AttributeError):
                continue
    if fn is None:
        return -1
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    fn.restype = C.c_int
    return fn(dst, src, nbytes, 2)


class Cloud:
    """Device-resident cloud handle (tc_cloud_*): owns a copy of the points in HBM, is indexed once, keeps its
    normals in the layout the ICP kernels read.  `points`: numpy (uploaded) or a torch CUDA tensor (copied on the device).

        prev = tc.Cloud(ctx, frame0); prev.estimate_normals(16, out=False)
        cur = tc.Cloud(ctx, frame1)
        r = cur.icp_point_to_plane(prev)          # icp_point_to_plane(source=cur, target=prev, prev's normals)
    """

    def __init__(self, ctx: "GpuContext", points):
        self._ctx, self._L = ctx, _lib.load()
        h = C.c_void_p()
        if _is_torch(points):
            import torch
            x = points.detach().to(torch.float32).contiguous().reshape(-1, 3)
            ctx._order(x.device)
            ctx._check(self._L.tc_cloud_upload_device(ctx._h, x.data_ptr(), x.shape[0], C.byref(h)))
            # `x` may be a temporary, or be overwritten by the caller's next torch op: torch's stream waits for the copy (an
            # event, no host wait -- the constructor used to synchronise: 0.1 ms per 1 M-point pair)
            ctx._release(x.device)
            self._torch_device = x.device
        else:
            x = _as_host(points)
            ctx._check(self._L.tc_cloud_upload(ctx._h, x.ctypes.data, x.shape[0], C.byref(h)))
            self._torch_device = None
        self._h = h

    def __len__(self):
        return int(self._L.tc_cloud_size(self._h))

    def estimate_normals(self, k: int = 10, config: "NormalEstimationConfig" = None, out=True):
        """estimate_normals_with_config (normals.rs:257-357) on the handle's cloud; the normals stay with the handle.
        out=True: also return the (n, 6) NormalPoint3f array (torch on the device for a torch-made handle, numpy otherwise);
        out=False: nothing is returned and the scattered 24-byte record stores are skipped."""
        c = self._ctx._cfg(config or NormalEstimationConfig(k_neighbors=k))
        n = len(self)
        if not out:
            self._ctx._check(self._L.tc_cloud_estimate_normals_device(self._h, C.byref(c), None))
            return None
        if self._torch_device is not None:
            import torch
            o = torch.empty((n, 6), dtype=torch.float32, device=self._torch_device)
            self._ctx._order(o.device)
            self._ctx._check(self._L.tc_cloud_estimate_normals_device(self._h, C.byref(c), o.data_ptr()))
            return o
        o = np.empty((n, 6), np.float32)
        self._ctx._check(self._L.tc_cloud_estimate_normals(self._h, C.byref(c), o.ctypes.data))
        return o

    def normals(self):
        """the handle's (n, 6) NormalPoint3f array in input order (tc_cloud_normals_device; made from the cell-sorted normals on
        demand), as a numpy copy; None when the handle has no normals"""
        p = self._L.tc_cloud_normals_device(self._h)
        if not p:
            return None
        n = len(self)
        self._ctx._check(self._L.tc_synchronize(self._ctx._h))
        host = np.empty((n, 6), np.float32)
        rc = _hip_memcpy_dtoh(host.ctypes.data, p, n * 24)
        if rc != 0:
            raise GpuError(f"hipMemcpy failed ({rc})")
        return host

    def set_normals(self, normals):
        """normals computed elsewhere: (n, 3), or the (n, 6) NormalPoint3f array of an estimate_normals call"""
        import torch
        if not _is_torch(normals):
            normals = torch.from_numpy(np.ascontiguousarray(np.asarray(normals, np.float32))).to(torch.device("cuda", self._ctx.device))
        t = normals.detach().to(torch.float32).contiguous()
        stride = 6 if (t.dim() == 2 and t.shape[1] == 6) else 3
        nn = t.shape[0] if t.dim() == 2 else t.numel() // 3
        self._ctx._order(t.device)
        self._ctx._check(self._L.tc_cloud_set_normals_device(self._h, t.data_ptr() + (12 if stride == 6 else 0), nn, stride))

    def _icp(self, fn, target, init, max_iters, max_correspondence_distance, convergence_threshold, correspondences, needs_normals):
        import torch
        ctx = self._ctx
        md = ctx._max_dist(max_correspondence_distance)
        if md is None:
            ctx._reject_all(len(self), len(target), max_iters)
        i7 = np.ascontiguousarray(IDENTITY if init is None else np.asarray(init, np.float32).reshape(7))
        r = _lib.IcpResultC()
        corr = None
        if correspondences:
            corr = torch.empty(max(1, len(self)), dtype=torch.int32, device=torch.device("cuda", ctx.device))
            r.corr_target = corr.data_ptr()
            ctx._order(corr.device)
        ctx._check(fn(self._h, target._h, i7.ctypes.data, max_iters, md, convergence_threshold, C.byref(r)))
        if corr is not None:
            corr = corr[: len(self)] if correspondences == "device" else corr[: len(self)].to(torch.int64) & 0xFFFFFFFF
        return ctx._result(r, len(self), corr, correspondences)

    def icp_point_to_plane(self, target: "Cloud", init=None, max_iters=50, max_correspondence_distance=None,
                           convergence_threshold=1e-6, correspondences=False):
        """icp_point_to_plane_detailed (registration.rs:508-602): self = source, `target` = a handle with normals"""
        return self._icp(self._L.tc_cloud_icp_point_to_plane, target, init, max_iters, max_correspondence_distance, convergence_threshold,
                         correspondences, True)

    def icp_detailed(self, target: "Cloud", init=None, max_iters=50, max_correspondence_distance=None, convergence_threshold=1e-6,
                     correspondences=False):
        """icp_detailed (registration.rs:258-370): self = source"""
        return self._icp(self._L.tc_cloud_icp_detailed, target, init, max_iters, max_correspondence_distance, convergence_threshold,
                         correspondences, False)

    def close(self):
        if getattr(self, "_h", None):
            self._L.tc_cloud_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- module-level free functions with the reference's names ----------------------------------
_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = GpuContext(0)
    return _default_ctx


def estimate_normals(cloud, k=10, ctx=None):
    return (ctx or default_context()).estimate_normals(cloud, k)


def estimate_normals_with_config(cloud, config, ctx=None):
    return (ctx or default_context()).estimate_normals_with_config(cloud, config)


def estimate_normals_radius(cloud, radius, consistent_orientation, ctx=None):
    return (ctx or default_context()).estimate_normals_radius(cloud, radius, consistent_orientation)


def voxel_grid_filter(cloud, voxel_size, ctx=None):
    return (ctx or default_context()).voxel_grid_filter(cloud, voxel_size)


def gpu_voxel_grid_filter(gpu_context, cloud, voxel_size):
    """threecrate-gpu/src/lib.rs:50 facade name; semantics of the CPU voxel_grid_filter (centroids)."""
    return gpu_context.voxel_grid_filter(cloud, voxel_size)


def icp(source, target, init=None, max_iters=50, ctx=None):
    return (ctx or default_context()).icp(source, target, init, max_iters)


def icp_detailed(source, target, init, max_iters, max_correspondence_distance=None, convergence_threshold=1e-6, ctx=None):
    return (ctx or default_context()).icp_detailed(source, target, init, max_iters, max_correspondence_distance,
                                                   convergence_threshold)


def icp_point_to_point(source, target, init, max_iterations, convergence_threshold=1e-6,
                       max_correspondence_distance=None, ctx=None):
    return (ctx or default_context()).icp_point_to_point(source, target, init, max_iterations, convergence_threshold,
                                                         max_correspondence_distance)


def gicp(source, target, init, config=None, ctx=None):
    """gicp.rs:100-105"""
    return (ctx or default_context()).gicp(source, target, init, config)


def kiss_icp(source, target, init, config=None, ctx=None):
    """kiss_icp.rs:183-188"""
    return (ctx or default_context()).kiss_icp(source, target, init, config)


def icp_point_to_point_default(source, target, init, max_iterations, ctx=None):
    """registration.rs:694-701"""
    return icp_point_to_point(source, target, init, max_iterations, 1e-6, None, ctx)


def icp_point_to_plane(source, target, target_normals, init, max_iters, ctx=None):
    return (ctx or default_context()).icp_point_to_plane(source, target, target_normals, init, max_iters)


def icp_point_to_plane_detailed(source, target, target_normals, init, max_iters, max_correspondence_distance=None,
                                convergence_threshold=1e-6, ctx=None):
    return (ctx or default_context()).icp_point_to_plane_detailed(source, target, target_normals, init, max_iters,
                                                                  max_correspondence_distance, convergence_threshold)


# ---- threecrate-gpu facade (threecrate-gpu/src/lib.rs:48-60) ---------------------------------
def gpu_estimate_normals(gpu_context, cloud, k):
    """threecrate-gpu/src/normals.rs:443-461"""
    return gpu_context.estimate_normals(cloud, k)


def gpu_icp(gpu_context, source, target, max_iterations, convergence_threshold, max_correspondence_distance):
    """threecrate-gpu/src/icp.rs:977-994 -> Isometry3 (7 floats); starts from identity (:202)."""
    return gpu_context.icp_point_to_point(source, target, None, max_iterations, convergence_threshold,
                                          max_correspondence_distance, correspondences=False).transformation


def gpu_icp_point_to_plane(gpu_context, source, target, target_normals, max_iterations, convergence_threshold,
                           max_correspondence_distance):
    """threecrate-gpu/src/icp.rs:1017-1036"""
    return gpu_context.icp_point_to_plane_detailed(source, target, target_normals, None, max_iterations,
                                                   max_correspondence_distance, convergence_threshold, correspondences=False)


@dataclass
class BatchICPJob:
    """threecrate-gpu/src/icp.rs:132-139"""
    source: np.ndarray
    target: np.ndarray
    max_iterations: int
    convergence_threshold: float
    max_correspondence_distance: float


@dataclass
class BatchICPResult:
    """threecrate-gpu/src/icp.rs:142-147 (+ per-job status)"""
    transformation: np.ndarray
    final_error: float
    iterations: int
    status: int = 0


def gpu_batch_icp(gpu_contexts, jobs):
    """threecrate-gpu/src/icp.rs:997-1002; job i runs on gpu_contexts[i % len]; contexts on
    different GPUs run concurrently."""
    ctxs = gpu_contexts if isinstance(gpu_contexts, (list, tuple)) else [gpu_contexts]
    L = _lib.load()
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    keep = []
    cj = (_lib.BatchJobC * max(1, len(jobs)))()
    for i, j in enumerate(jobs):
        s, t = _as_host(j.source), _as_host(j.target)
        keep += [s, t]
        cj[i].source, cj[i].n_source, cj[i].target, cj[i].n_target = s.ctypes.data, s.shape[0], t.ctypes.data, t.shape[0]
        cj[i].max_iterations = j.max_iterations
        cj[i].convergence_threshold = j.convergence_threshold
        cj[i].max_correspondence_distance = j.max_correspondence_distance
    cr = (_lib.BatchResultC * max(1, len(jobs)))()
    rc = L.tc_batch_icp(arr, len(ctxs), cj, len(jobs), cr)
    if rc != _lib.TC_OK:
        raise InvalidData("tc_batch_icp: bad arguments")
    return [BatchICPResult(np.array(list(cr[i].transformation), np.float32), float(cr[i].final_error),
                           int(cr[i].iterations), int(cr[i].status)) for i in range(len(jobs))]


# ---- LiDAR frame streaming (RealtimePipeline, threecrate-algorithms/src/streaming.rs:540-646) -------
@dataclass
class BackpressureConfig:
    """streaming.rs BackpressureConfig: only max_queue_depth applies to whole-frame items."""
    max_queue_depth: int = 4


@dataclass
class FrameResult:
    transformation: np.ndarray      # current frame -> previous frame (7-float Isometry3)
    mse: float
    iterations: int
    converged: bool
    status: int
    n_points_in: int
    n_points: int

    @property
    def matrix(self):
        return isometry_to_matrix(self.transformation)


@dataclass
class RealtimeMetrics:
    items_queued: int
    items_processed: int
    items_dropped: int
    max_depth_seen: int


class FrameStream:
    """Bounded-queue frame registration pipeline on one GPU (tc_frame_stream_*): send() blocks when the
    queue is full, try_send() drops instead, finish() drains and returns (results, metrics).  Frames are
    host arrays (n, 3) or KITTI records (n, 4)."""

    def __init__(self, ctx: "GpuContext", max_points: int, voxel_size: float = 0.2, k_neighbors: int = 16,
                 max_iterations: int = 50, max_correspondence_distance=None, convergence_threshold: float = 1e-6,
                 backpressure: BackpressureConfig = None):
        self._ctx, self._L = ctx, _lib.load()
        bp = backpressure or BackpressureConfig()
        cfg = _lib.FrameStreamConfigC(max_points, bp.max_queue_depth, voxel_size, k_neighbors, max_iterations,
                                      -1.0 if max_correspondence_distance is None else float(max_correspondence_distance),
                                      convergence_threshold)
        h = C.c_void_p()
        ctx._check(self._L.tc_frame_stream_create(ctx._h, C.byref(cfg), C.byref(h)))
        self._h = h
        self._sent = 0

    def _frame(self, frame):
        a = np.ascontiguousarray(np.asarray(frame, np.float32))
        if a.ndim != 2 or a.shape[1] not in (3, 4):
            raise InvalidData("a frame is an (n, 3) xyz or (n, 4) KITTI x, y, z, intensity array")
        return a

    def send(self, frame):
        a = self._frame(frame)
        self._ctx._check(self._L.tc_frame_stream_send(self._h, a.ctypes.data, a.shape[0], a.shape[1]))
        self._sent += 1

    def try_send(self, frame) -> bool:
        a = self._frame(frame)
        ok = C.c_int(0)
        self._ctx._check(self._L.tc_frame_stream_try_send(self._h, a.ctypes.data, a.shape[0], a.shape[1], C.byref(ok)))
        self._sent += int(ok.value)
        return bool(ok.value)

    def finish(self):
        cap = max(self._sent, 1)
        res = (_lib.FrameResultC * cap)()
        n = C.c_size_t(0)
        m = _lib.FrameStreamMetricsC()
        rc = self._L.tc_frame_stream_finish(self._h, res, cap, C.byref(n), C.byref(m))
        self._L.tc_frame_stream_destroy(self._h)
        self._h = None
        self._ctx._check(rc)
        out = [FrameResult(np.array(list(r.transformation), np.float32), float(r.mse), int(r.iterations), bool(r.converged),
                           int(r.status), int(r.n_points_in), int(r.n_points)) for r in res[:min(n.value, cap)]]
        return out, RealtimeMetrics(int(m.items_queued), int(m.items_processed), int(m.items_dropped), int(m.max_depth_seen))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.tc_frame_stream_destroy(self._h)
                self._h = None
        except Exception:
            pass


class SearchIndex:
    """Persistent neighbour-search object (tc_search_index_*): KdTree::new once (nearest_neighbor.rs:37-58), then
    find_k_nearest / find_radius_neighbors (core/traits.rs:6-12) for any number of queries against the same cloud.
    The cloud (numpy or torch-on-device) is copied, cell-sorted, into device memory owned by the handle."""

    def __init__(self, ctx: "GpuContext", cloud, k_hint: int = 16):
        self._ctx, self._L = ctx, _lib.load()
        h = C.c_void_p()
        if _is_torch(cloud):
            import torch
            x = cloud.detach().to(torch.float32).contiguous().reshape(-1, 3)
            ctx._order(x.device)
            ctx._check(self._L.tc_search_index_create_device(ctx._h, x.data_ptr(), x.shape[0], int(k_hint), C.byref(h)))
        else:
            x = _as_host(cloud)
            ctx._check(self._L.tc_search_index_create(ctx._h, x.ctypes.data, x.shape[0], int(k_hint), C.byref(h)))
        self._h = h

    def __len__(self):
        return int(self._L.tc_search_index_size(self._h))

    def _query(self, queries, k, radius):
        kk = max(int(k), 1)
        if _is_torch(queries):
            import torch
            q = queries.detach().to(torch.float32).contiguous().reshape(-1, 3)
            idx = torch.zeros((q.shape[0], kk), dtype=torch.int32, device=q.device)
            dist = torch.zeros((q.shape[0], kk), dtype=torch.float32, device=q.device)
            cnt = torch.zeros(q.shape[0], dtype=torch.int32, device=q.device)
            self._ctx._order(q.device)
            self._ctx._check(self._L.tc_search_index_query_device(self._h, q.data_ptr(), q.shape[0], int(k), float(radius), idx.data_ptr(),
                                                                  dist.data_ptr(), cnt.data_ptr()))
            return idx, dist, cnt
        q = _as_host(queries)
        idx = np.zeros((len(q), kk), np.uint32)
        dist = np.zeros((len(q), kk), np.float32)
        cnt = np.zeros(len(q), np.uint32)
        self._ctx._check(self._L.tc_search_index_query(self._h, q.ctypes.data, q.shape[0], int(k), float(radius), idx.ctypes.data,
                                                       dist.ctypes.data, cnt.ctypes.data))
        return idx.astype(np.int64), dist, cnt

    def find_k_nearest_batch(self, queries, k: int):
        """(idx (nq, k), dist (nq, k), count (nq,)); rows ascending by distance, entries past count[q] undefined"""
        return self._query(queries, k, -1.0)

    def find_radius_neighbors_batch(self, queries, radius: float, k_max: int = 32):
        """the neighbours within radius among the k_max nearest; count[q] == k_max: there may be more"""
        return self._query(queries, k_max, max(float(radius), 0.0))

    def radius_counts(self, queries, radius: float):
        """number of cloud points within `radius` of every host query (tc_search_index_radius_count)"""
        q = _as_host(queries)
        cnt = np.zeros(len(q), np.uint32)
        self._ctx._check(self._L.tc_search_index_radius_count(self._h, q.ctypes.data, q.shape[0], float(radius), cnt.ctypes.data))
        return cnt

    def find_radius_neighbors_all(self, queries, radius: float):
        """NearestNeighborSearch::find_radius_neighbors (nearest_neighbor.rs:254-298) for many host queries, WITHOUT a cap:
        (offsets (nq + 1,) int64, idx (total,) int64, dist (total,) f32); query q owns [offsets[q], offsets[q + 1]),
        ascending by distance like the reference's final sort."""
        q = _as_host(queries)
        cnt = np.zeros(len(q), np.uint32)
        self._ctx._check(self._L.tc_search_index_radius_count(self._h, q.ctypes.data, q.shape[0], float(radius), cnt.ctypes.data))
        off = np.zeros(len(q) + 1, np.uint64)
        np.cumsum(cnt, out=off[1:])
        total = int(off[-1])
        idx, dist = np.zeros(total, np.uint32), np.zeros(total, np.float32)
        if total:
            self._ctx._check(self._L.tc_search_index_radius_fill(self._h, q.ctypes.data, q.shape[0], float(radius), off.ctypes.data, total,
                                                                 idx.ctypes.data, dist.ctypes.data))
            seg = np.repeat(np.arange(len(q)), cnt.astype(np.int64))
            order = np.lexsort((idx, dist, seg))            # per query: by distance, ties by index
            idx, dist = idx[order], dist[order]
        return off.astype(np.int64), idx.astype(np.int64), dist

    def find_k_nearest(self, query, k: int):
        idx, dist, cnt = self.find_k_nearest_batch(np.asarray(query, np.float32).reshape(1, 3), k)
        return [(int(idx[0, i]), float(dist[0, i])) for i in range(int(cnt[0]))]

    def find_radius_neighbors(self, query, radius: float, k_max: int = 32):
        idx, dist, cnt = self.find_radius_neighbors_batch(np.asarray(query, np.float32).reshape(1, 3), radius, k_max)
        return [(int(idx[0, i]), float(dist[0, i])) for i in range(int(cnt[0]))]

    def close(self):
        if self._h:
            self._L.tc_search_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_kitti_bin(path):
    """VelodyneKittiBinReader::read (threecrate-io/src/lidar.rs:310-343) -> (n, 3) float32."""
    L = _lib.load()
    n = C.c_size_t(0)
    p = os.fsencode(path)
    rc = L.tc_read_kitti_bin(p, None, 0, C.byref(n))
    if rc != _lib.TC_OK:
        raise InvalidData(f"Velodyne KITTI binary: cannot read {path} or its size is not a multiple of 16 bytes")
    out = np.empty((n.value, 3), np.float32)
    if n.value:
        rc = L.tc_read_kitti_bin(p, out.ctypes.data, n.value, C.byref(n))
        if rc != _lib.TC_OK:
            raise InvalidData(f"Velodyne KITTI binary: cannot read {path}")
    return out
