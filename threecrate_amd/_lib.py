"""ctypes binding of libthreecrate_hip.so (include/threecrate_hip.h).

The shared library is the product; this module only declares its C ABI.  There is no
Python / CPU fallback: if the library is missing, `load()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TC_HIP_LIB: another build of the same library (the host-side ASan / UBSan build of tools/sanitize_cpu.sh)
LIB_PATH = os.environ.get("TC_HIP_LIB") or os.path.join(_HERE, "libthreecrate_hip.so")

TC_OK, TC_INVALID_DATA, TC_ALGORITHM, TC_GPU, TC_UNSUPPORTED = 0, 1, 2, 3, 4
TC_COMM_ID_BYTES = 128
TC_COLL_SUM_F64, TC_COLL_SUM_U32, TC_COLL_ALLGATHER_U8 = 0, 1, 2
TC_SHARD_SPATIAL, TC_SHARD_LOCAL, TC_SHARD_INDEX = 0, 1, 2
TC_COUNTER_INDEXED_POINTS, TC_COUNTER_INDEX_BUILDS = 0, 1
(TC_COUNTER_ICP_ITERATIONS, TC_COUNTER_ICP_TRIPS, TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH, TC_COUNTER_ICP_SEARCHES, TC_COUNTER_ICP_STEPS_NEEDED,
 TC_COUNTER_ICP_STEPS_TAKEN) = 2, 3, 4, 5, 6, 7
# int (*tc_host_collective_fn)(void *user, int op, void *host_buf, size_t count)
HOST_COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
SUMS_P2PLANE, SUMS_P2P, SUMS_STRIDE = 29, 17, 32


class NormalConfig(C.Structure):
    _fields_ = [("k_neighbors", C.c_uint64), ("radius", C.c_float), ("has_radius", C.c_int32),
                ("consistent_orientation", C.c_int32), ("has_viewpoint", C.c_int32),
                ("viewpoint", C.c_float * 3)]


class IcpResultC(C.Structure):
    _fields_ = [("transformation", C.c_float * 7), ("mse", C.c_float), ("iterations", C.c_uint64),
                ("converged", C.c_int32), ("n_correspondences", C.c_uint64),
                ("corr_target", C.c_void_p)]


class BatchJobC(C.Structure):
    _fields_ = [("source", C.c_void_p), ("n_source", C.c_size_t), ("target", C.c_void_p),
                ("n_target", C.c_size_t), ("max_iterations", C.c_size_t),
                ("convergence_threshold", C.c_float), ("max_correspondence_distance", C.c_float)]


class BatchResultC(C.Structure):
    _fields_ = [("transformation", C.c_float * 7), ("final_error", C.c_float),
                ("iterations", C.c_uint64), ("status", C.c_int32)]


class ScaleLevelC(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("max_iterations", C.c_size_t), ("max_correspondence_distance", C.c_float)]


class MultiScaleConfigC(C.Structure):
    _fields_ = [("levels", C.POINTER(ScaleLevelC)), ("n_levels", C.c_size_t), ("final_refinement_iterations", C.c_size_t),
                ("final_max_correspondence_distance", C.c_float), ("convergence_threshold", C.c_float)]


class GicpConfigC(C.Structure):
    _fields_ = [("max_iterations", C.c_size_t), ("max_correspondence_distance", C.c_float), ("convergence_threshold", C.c_float),
                ("k_correspondences", C.c_size_t)]


class KissIcpConfigC(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("max_range", C.c_float), ("min_range", C.c_float), ("max_iterations", C.c_size_t)]


class FrameStreamConfigC(C.Structure):
    _fields_ = [("max_points", C.c_size_t), ("max_queue_depth", C.c_size_t), ("voxel_size", C.c_float),
                ("k_neighbors", C.c_size_t), ("max_iterations", C.c_size_t),
                ("max_correspondence_distance", C.c_float), ("convergence_threshold", C.c_float)]


class FrameResultC(C.Structure):
    _fields_ = [("transformation", C.c_float * 7), ("mse", C.c_float), ("iterations", C.c_uint64),
                ("converged", C.c_int32), ("status", C.c_int32), ("n_points_in", C.c_uint64), ("n_points", C.c_uint64)]


class FrameStreamMetricsC(C.Structure):
    _fields_ = [("items_queued", C.c_uint64), ("items_processed", C.c_uint64), ("items_dropped", C.c_uint64),
                ("max_depth_seen", C.c_uint64)]


class KernelStatC(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("min_ms", C.c_double), ("max_ms", C.c_double)]


# every symbol include/threecrate_hip.h declares (tests/test_abi_symbols.py checks the header against this)
EXPORTS = [
    "tc_abi_version", "tc_device_count", "tc_context_create", "tc_context_create_on_stream", "tc_context_wait_stream", "tc_stream_wait_context",
    "tc_context_trim", "tc_context_destroy", "tc_last_error_message", "tc_synchronize", "tc_normal_config_default",
    "tc_estimate_normals", "tc_estimate_normals_device", "tc_estimate_normals_slice_device", "tc_normals_unsort_device", "tc_icp_detailed", "tc_icp_detailed_device",
    "tc_icp_point_to_point", "tc_icp", "tc_icp_point_to_plane_detailed",
    "tc_icp_point_to_plane_detailed_device", "tc_batch_icp", "tc_icp_shard_create", "tc_icp_shard_sums",
    "tc_icp_shard_reduce", "tc_icp_shard_get_sums", "tc_icp_shard_set_sums", "tc_icp_shard_done",
    "tc_icp_shard_apply", "tc_icp_shard_finish", "tc_icp_shard_destroy",
    "tc_cloud_upload", "tc_cloud_upload_device", "tc_cloud_size", "tc_cloud_points_device", "tc_cloud_normals_device",
    "tc_cloud_estimate_normals", "tc_cloud_estimate_normals_device", "tc_cloud_set_normals_device", "tc_cloud_icp_point_to_plane",
    "tc_cloud_icp_detailed", "tc_cloud_sharded_icp", "tc_cloud_destroy",
    "tc_comm_unique_id", "tc_comm_create", "tc_comm_adopt", "tc_comm_create_host", "tc_comm_create_local", "tc_comm_rank", "tc_comm_size",
    "tc_comm_destroy", "tc_sharded_icp_point_to_plane_device", "tc_sharded_icp_detailed_device", "tc_sharded_estimate_normals_device",
    "tc_sharded_estimate_normals_local_device", "tc_debug_counter",
    "tc_multiscale_icp_point_to_point", "tc_gicp", "tc_gicp_device", "tc_kiss_icp", "tc_kiss_icp_device", "tc_knn", "tc_knn_device", "tc_radius_search", "tc_radius_search_device",
    "tc_search_index_create", "tc_search_index_create_device", "tc_search_index_size", "tc_search_index_query",
    "tc_search_index_query_device", "tc_search_index_radius_count", "tc_search_index_radius_fill", "tc_search_index_destroy", "tc_voxel_grid_filter", "tc_voxel_grid_filter_device",
    "tc_frame_stream_create", "tc_frame_stream_send", "tc_frame_stream_try_send", "tc_frame_stream_finish",
    "tc_frame_stream_destroy", "tc_read_kitti_bin", "tc_profile_enable", "tc_profile_reset", "tc_profile_read",
]

_lib = None


def _preload_hip_runtime():
    """If PyTorch-ROCm is installed it bundles its own libamdhip64.so (SONAME libamdhip64.so.7).
    Two HIP runtimes in one process cannot both open the GPU, so bind to torch's copy first:
    our NEEDED libamdhip64.so.7 then resolves to it and torch tensors / streams can be handed
    across the C ABI.  Without torch the system ROCm runtime (/opt/rocm/lib) is used."""
    if os.environ.get("TC_NO_TORCH_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  threecrate_amd has no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, f32p, sz, f, i = C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_int
    ctxpp = C.POINTER(C.c_void_p)
    resp = C.POINTER(IcpResultC)
    L.tc_abi_version.restype = i
    L.tc_device_count.restype = i
    L.tc_context_create.argtypes = [i, ctxpp]
    L.tc_context_create_on_stream.argtypes = [i, vp, ctxpp]
    L.tc_context_wait_stream.argtypes = [vp, vp]
    L.tc_stream_wait_context.argtypes = [vp, vp]
    L.tc_context_trim.argtypes = [vp]
    L.tc_context_trim.restype = C.c_int
    L.tc_context_destroy.argtypes = [vp]
    L.tc_context_destroy.restype = None
    L.tc_last_error_message.argtypes = [vp]
    L.tc_last_error_message.restype = C.c_char_p
    L.tc_synchronize.argtypes = [vp]
    L.tc_normal_config_default.argtypes = [C.POINTER(NormalConfig)]
    L.tc_normal_config_default.restype = None
    L.tc_estimate_normals.argtypes = [vp, f32p, sz, C.POINTER(NormalConfig), f32p]
    L.tc_estimate_normals_device.argtypes = [vp, f32p, sz, C.POINTER(NormalConfig), f32p]
    L.tc_estimate_normals_slice_device.argtypes = [vp, f32p, sz, C.POINTER(NormalConfig), sz, sz, f32p]
    L.tc_normals_unsort_device.argtypes = [vp, f32p, sz, f32p]
    L.tc_icp_detailed.argtypes = [vp, f32p, sz, f32p, sz, f32p, sz, f, f, resp]
    L.tc_icp_detailed_device.argtypes = [vp, f32p, sz, f32p, sz, f32p, sz, f, f, resp]
    L.tc_icp_point_to_point.argtypes = [vp, f32p, sz, f32p, sz, f32p, sz, f, f, resp]
    L.tc_icp.argtypes = [vp, f32p, sz, f32p, sz, f32p, sz, f32p]
    L.tc_icp_point_to_plane_detailed.argtypes = [vp, f32p, sz, f32p, sz, f32p, sz, sz, f32p, sz, f, f, resp]
    L.tc_icp_point_to_plane_detailed_device.argtypes = [vp, f32p, sz, f32p, sz, f32p, sz, sz, f32p, sz, f, f, resp]
    L.tc_batch_icp.argtypes = [ctxpp, sz, C.POINTER(BatchJobC), sz, C.POINTER(BatchResultC)]
    L.tc_icp_shard_create.argtypes = [vp, i, f32p, sz, f32p, sz, f32p, sz, f32p, f, f, ctxpp]
    L.tc_icp_shard_sums.argtypes = [vp]
    L.tc_icp_shard_sums.restype = C.c_void_p
    L.tc_icp_shard_reduce.argtypes = [vp]
    L.tc_icp_shard_get_sums.argtypes = [vp, vp]
    L.tc_icp_shard_set_sums.argtypes = [vp, vp]
    L.tc_icp_shard_done.argtypes = [vp, C.POINTER(C.c_int)]
    L.tc_icp_shard_apply.argtypes = [vp]
    L.tc_icp_shard_finish.argtypes = [vp, sz, resp]
    L.tc_icp_shard_destroy.argtypes = [vp]
    L.tc_icp_shard_destroy.restype = None
    L.tc_cloud_upload.argtypes = [vp, f32p, sz, ctxpp]
    L.tc_cloud_upload_device.argtypes = [vp, f32p, sz, ctxpp]
    L.tc_cloud_size.argtypes = [vp]
    L.tc_cloud_size.restype = sz
    L.tc_cloud_points_device.argtypes = [vp]
    L.tc_cloud_points_device.restype = C.c_void_p
    L.tc_cloud_normals_device.argtypes = [vp]
    L.tc_cloud_normals_device.restype = C.c_void_p
    L.tc_cloud_estimate_normals.argtypes = [vp, C.POINTER(NormalConfig), f32p]
    L.tc_cloud_estimate_normals_device.argtypes = [vp, C.POINTER(NormalConfig), f32p]
    L.tc_cloud_set_normals_device.argtypes = [vp, f32p, sz, sz]
    L.tc_cloud_icp_point_to_plane.argtypes = [vp, vp, f32p, sz, f, f, resp]
    L.tc_cloud_icp_detailed.argtypes = [vp, vp, f32p, sz, f, f, resp]
    L.tc_cloud_sharded_icp.argtypes = [vp, i, i, f32p, sz, vp, f32p, sz, f, f, resp]
    L.tc_cloud_destroy.argtypes = [vp]
    L.tc_cloud_destroy.restype = None
    L.tc_comm_unique_id.argtypes = [vp]
    L.tc_comm_create.argtypes = [vp, i, i, vp, ctxpp]
    L.tc_comm_adopt.argtypes = [vp, vp, i, i, ctxpp]
    L.tc_comm_create_host.argtypes = [vp, i, i, HOST_COLLECTIVE_FN, vp, ctxpp]
    L.tc_comm_create_local.argtypes = [vp, ctxpp]
    L.tc_comm_rank.argtypes = [vp]
    L.tc_comm_rank.restype = i
    L.tc_comm_size.argtypes = [vp]
    L.tc_comm_size.restype = i
    L.tc_comm_destroy.argtypes = [vp]
    L.tc_comm_destroy.restype = None
    L.tc_sharded_icp_point_to_plane_device.argtypes = [vp, vp, i, f32p, sz, f32p, sz, f32p, sz, sz, f32p, sz, f, f, resp]
    L.tc_sharded_icp_detailed_device.argtypes = [vp, vp, i, f32p, sz, f32p, sz, f32p, sz, f, f, resp]
    L.tc_sharded_estimate_normals_device.argtypes = [vp, vp, f32p, sz, C.POINTER(NormalConfig), f32p]
    L.tc_sharded_estimate_normals_local_device.argtypes = [vp, vp, f32p, sz, C.POINTER(NormalConfig), f32p, vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.tc_debug_counter.argtypes = [vp, i]
    L.tc_debug_counter.restype = C.c_ulonglong
    L.tc_multiscale_icp_point_to_point.argtypes = [vp, f32p, sz, f32p, sz, f32p, C.POINTER(MultiScaleConfigC), resp]
    L.tc_gicp.argtypes = [vp, f32p, sz, f32p, sz, f32p, C.POINTER(GicpConfigC), resp]
    L.tc_gicp_device.argtypes = [vp, f32p, sz, f32p, sz, f32p, C.POINTER(GicpConfigC), resp]
    L.tc_kiss_icp.argtypes = [vp, f32p, sz, f32p, sz, f32p, C.POINTER(KissIcpConfigC), resp, C.POINTER(C.c_size_t)]
    L.tc_kiss_icp_device.argtypes = [vp, f32p, sz, f32p, sz, f32p, C.POINTER(KissIcpConfigC), resp, C.POINTER(C.c_size_t)]
    L.tc_knn.argtypes = [vp, f32p, sz, f32p, sz, sz, vp, vp, vp]
    L.tc_knn_device.argtypes = [vp, f32p, sz, f32p, sz, sz, vp, vp, vp]
    L.tc_radius_search.argtypes = [vp, f32p, sz, f32p, sz, f, sz, vp, vp, vp]
    L.tc_radius_search_device.argtypes = [vp, f32p, sz, f32p, sz, f, sz, vp, vp, vp]
    L.tc_search_index_create.argtypes = [vp, f32p, sz, sz, ctxpp]
    L.tc_search_index_create_device.argtypes = [vp, f32p, sz, sz, ctxpp]
    L.tc_search_index_size.argtypes = [vp]
    L.tc_search_index_size.restype = sz
    L.tc_search_index_query.argtypes = [vp, f32p, sz, sz, f, vp, vp, vp]
    L.tc_search_index_query_device.argtypes = [vp, f32p, sz, sz, f, vp, vp, vp]
    L.tc_search_index_radius_count.argtypes = [vp, f32p, sz, f, vp]
    L.tc_search_index_radius_fill.argtypes = [vp, f32p, sz, f, vp, sz, vp, vp]
    L.tc_search_index_destroy.argtypes = [vp]
    L.tc_search_index_destroy.restype = None
    L.tc_voxel_grid_filter.argtypes = [vp, f32p, sz, f, f32p, C.POINTER(C.c_size_t)]
    L.tc_voxel_grid_filter_device.argtypes = [vp, f32p, sz, f, f32p, C.POINTER(C.c_size_t)]
    L.tc_frame_stream_create.argtypes = [vp, C.POINTER(FrameStreamConfigC), ctxpp]
    L.tc_frame_stream_send.argtypes = [vp, f32p, sz, sz]
    L.tc_frame_stream_try_send.argtypes = [vp, f32p, sz, sz, C.POINTER(C.c_int)]
    L.tc_frame_stream_finish.argtypes = [vp, C.POINTER(FrameResultC), sz, C.POINTER(C.c_size_t), C.POINTER(FrameStreamMetricsC)]
    L.tc_frame_stream_destroy.argtypes = [vp]
    L.tc_frame_stream_destroy.restype = None
    L.tc_read_kitti_bin.argtypes = [C.c_char_p, f32p, sz, C.POINTER(C.c_size_t)]
    L.tc_profile_enable.argtypes = [vp, i]
    L.tc_profile_enable.restype = None
    L.tc_profile_reset.argtypes = [vp]
    L.tc_profile_reset.restype = None
    L.tc_profile_read.argtypes = [vp, C.POINTER(KernelStatC), sz]
    L.tc_profile_read.restype = sz
    _lib = L
    return L
