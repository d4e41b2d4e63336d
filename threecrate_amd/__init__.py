"""threecrate_amd -- MI355X (gfx950) native normals + ICP backend for threecrate.

The product is libthreecrate_hip.so (hand-written HIP kernels behind the C ABI of
include/threecrate_hip.h).  This package is the thin host-side mirror of the reference's
operator interface for that one path (estimate_normals / icp / icp_point_to_plane and the
threecrate-gpu facade).  It never falls back to a CPU implementation.
"""
from .api import (  # noqa: F401
    IcpScaleLevel, MultiScaleIcpConfig,
    AlgorithmError, BatchICPJob, BatchICPResult, Error, GpuContext, GpuError, ICPResult, IDENTITY, InvalidData,
    NormalEstimationConfig, Unsupported, default_context, estimate_normals, estimate_normals_radius,
    estimate_normals_with_config, gpu_batch_icp, gpu_estimate_normals, gpu_icp, gpu_icp_point_to_plane, icp,
    icp_detailed, icp_point_to_plane, icp_point_to_plane_detailed, icp_point_to_point, icp_point_to_point_default,
    isometry_to_matrix, voxel_grid_filter, gpu_voxel_grid_filter,
    GicpConfig, gicp, KissIcpConfig, kiss_icp, BackpressureConfig, FrameResult, FrameStream, RealtimeMetrics, read_kitti_bin, SearchIndex, Cloud,
)

__all__ = [n for n in dir() if not n.startswith("_")]
