//! Raw declarations of include/threecrate_hip.h (keep in sync with the header; tc_abi_version() == 2).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct tc_context { _private: [u8; 0] }
#[repr(C)] pub struct tc_frame_stream { _private: [u8; 0] }
#[repr(C)] pub struct tc_search_index { _private: [u8; 0] }
#[repr(C)] pub struct tc_comm { _private: [u8; 0] }
#[repr(C)] pub struct tc_cloud { _private: [u8; 0] }

pub const TC_COMM_ID_BYTES: usize = 128;
pub const TC_SHARD_SPATIAL: c_int = 0;
pub const TC_SHARD_LOCAL: c_int = 1;
pub const TC_SHARD_INDEX: c_int = 2;
pub const TC_COUNTER_INDEXED_POINTS: c_int = 0;
pub const TC_COUNTER_INDEX_BUILDS: c_int = 1;
pub const TC_COUNTER_ICP_ITERATIONS: c_int = 2;
pub const TC_COUNTER_ICP_TRIPS: c_int = 3;
pub const TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH: c_int = 4;
pub const TC_COUNTER_ICP_SEARCHES: c_int = 5;
pub const TC_COUNTER_ICP_STEPS_NEEDED: c_int = 6;
pub const TC_COUNTER_ICP_STEPS_TAKEN: c_int = 7;
pub const TC_COLL_SUM_F64: c_int = 0;
pub const TC_COLL_SUM_U32: c_int = 1;
pub const TC_COLL_ALLGATHER_U8: c_int = 2;
/// `int (*tc_host_collective_fn)(void *user, int op, void *host_buf, size_t count)`
pub type tc_host_collective_fn = Option<unsafe extern "C" fn(user: *mut c_void, op: c_int, host_buf: *mut c_void, count: usize) -> c_int>;

pub const TC_OK: c_int = 0;
pub const TC_INVALID_DATA: c_int = 1;
pub const TC_ALGORITHM: c_int = 2;
pub const TC_GPU: c_int = 3;
pub const TC_UNSUPPORTED: c_int = 4;

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_normal_config {                 // NormalEstimationConfig, normals.rs:17-37
    pub k_neighbors: u64,
    pub radius: f32,
    pub has_radius: i32,
    pub consistent_orientation: i32,
    pub has_viewpoint: i32,
    pub viewpoint: [f32; 3],
}

#[repr(C)] #[derive(Debug)]
pub struct tc_icp_result {                    // ICPResult, registration.rs:13-24
    pub transformation: [f32; 7],             // qi qj qk qw tx ty tz (nalgebra Isometry3 storage order)
    pub mse: f32,
    pub iterations: u64,
    pub converged: i32,
    pub n_correspondences: u64,
    pub corr_target: *mut u32,                // caller-allocated, n_source entries, u32::MAX = no match
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_batch_icp_job {                 // BatchICPJob, threecrate-gpu/src/icp.rs:132-139 (clouds as n x 3 host floats)
    pub source: *const f32, pub n_source: usize,
    pub target: *const f32, pub n_target: usize,
    pub max_iterations: usize,
    pub convergence_threshold: f32,
    pub max_correspondence_distance: f32,     // < 0: none
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_batch_icp_result {              // BatchICPResult, threecrate-gpu/src/icp.rs:142-147 (+ the job's tc_status)
    pub transformation: [f32; 7],
    pub final_error: f32,
    pub iterations: u64,
    pub status: i32,
}

#[repr(C)] #[derive(Clone, Copy)]
pub struct tc_kernel_stat { pub name: [c_char; 48], pub launches: u64, pub total_ms: f64, pub min_ms: f64, pub max_ms: f64 }

#[repr(C)] pub struct tc_icp_shard { _private: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_gicp_config { pub max_iterations: usize, pub max_correspondence_distance: f32, pub convergence_threshold: f32, pub k_correspondences: usize }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_kiss_icp_config { pub voxel_size: f32, pub max_range: f32, pub min_range: f32, pub max_iterations: usize }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_icp_scale_level { pub voxel_size: f32, pub max_iterations: usize, pub max_correspondence_distance: f32 }

#[repr(C)] #[derive(Debug)]
pub struct tc_multiscale_icp_config {
    pub levels: *const tc_icp_scale_level,
    pub n_levels: usize,
    pub final_refinement_iterations: usize,
    pub final_max_correspondence_distance: f32,
    pub convergence_threshold: f32,
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_frame_stream_config {
    pub max_points: usize, pub max_queue_depth: usize, pub voxel_size: f32, pub k_neighbors: usize,
    pub max_iterations: usize, pub max_correspondence_distance: f32, pub convergence_threshold: f32,
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct tc_frame_result {
    pub transformation: [f32; 7], pub mse: f32, pub iterations: u64, pub converged: i32, pub status: i32,
    pub n_points_in: u64, pub n_points: u64,
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct tc_frame_stream_metrics { pub items_queued: u64, pub items_processed: u64, pub items_dropped: u64, pub max_depth_seen: u64 }

extern "C" {
    pub fn tc_abi_version() -> c_int;
    pub fn tc_device_count() -> c_int;
    pub fn tc_context_create(device: c_int, out: *mut *mut tc_context) -> c_int;
    pub fn tc_context_create_on_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut tc_context) -> c_int;
    pub fn tc_context_trim(ctx: *mut tc_context) -> c_int;
    pub fn tc_context_destroy(ctx: *mut tc_context);
    pub fn tc_last_error_message(ctx: *const tc_context) -> *const c_char;
    pub fn tc_normal_config_default(cfg: *mut tc_normal_config);
    pub fn tc_estimate_normals(ctx: *mut tc_context, xyz: *const f32, n: usize, cfg: *const tc_normal_config, out: *mut f32) -> c_int;
    pub fn tc_icp_detailed(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize, init: *const f32,
                           max_iters: usize, max_dist: f32, conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_icp_point_to_point(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize, init: *const f32,
                                 max_iters: usize, conv_thr: f32, max_dist: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_icp(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize, init: *const f32, max_iters: usize,
                  out: *mut f32) -> c_int;
    pub fn tc_icp_point_to_plane_detailed(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize,
                                          normals: *const f32, n_normals: usize, stride: usize, init: *const f32, max_iters: usize,
                                          max_dist: f32, conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_multiscale_icp_point_to_point(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize,
                                            init: *const f32, cfg: *const tc_multiscale_icp_config, res: *mut tc_icp_result) -> c_int;
    pub fn tc_gicp(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize, init: *const f32,
                   cfg: *const tc_gicp_config, res: *mut tc_icp_result) -> c_int;
    pub fn tc_kiss_icp(ctx: *mut tc_context, src: *const f32, ns: usize, tgt: *const f32, nt: usize, init: *const f32,
                       cfg: *const tc_kiss_icp_config, res: *mut tc_icp_result, n_source_down: *mut usize) -> c_int;
    pub fn tc_knn(ctx: *mut tc_context, cloud: *const f32, n: usize, queries: *const f32, nq: usize, k: usize,
                  idx: *mut u32, dist: *mut f32, count: *mut u32) -> c_int;
    pub fn tc_radius_search(ctx: *mut tc_context, cloud: *const f32, n: usize, queries: *const f32, nq: usize, radius: f32, k_max: usize,
                            idx: *mut u32, dist: *mut f32, count: *mut u32) -> c_int;
    pub fn tc_estimate_normals_slice_device(ctx: *mut tc_context, d_xyz: *const f32, n: usize, cfg: *const tc_normal_config, begin: usize, end: usize,
                                            d_slice_out: *mut f32) -> c_int;
    pub fn tc_normals_unsort_device(ctx: *mut tc_context, d_sorted_all: *const f32, n: usize, d_out: *mut f32) -> c_int;
    pub fn tc_search_index_create(ctx: *mut tc_context, cloud: *const f32, n: usize, k_hint: usize, out: *mut *mut tc_search_index) -> c_int;
    pub fn tc_search_index_size(index: *const tc_search_index) -> usize;
    pub fn tc_search_index_query(index: *mut tc_search_index, queries: *const f32, nq: usize, k: usize, radius: f32,
                                 idx: *mut u32, dist: *mut f32, count: *mut u32) -> c_int;
    pub fn tc_search_index_radius_count(index: *mut tc_search_index, queries: *const f32, nq: usize, radius: f32, counts: *mut u32) -> c_int;
    pub fn tc_search_index_radius_fill(index: *mut tc_search_index, queries: *const f32, nq: usize, radius: f32, offsets: *const u64, total: usize,
                                       idx: *mut u32, dist: *mut f32) -> c_int;
    pub fn tc_search_index_destroy(index: *mut tc_search_index);
    pub fn tc_voxel_grid_filter(ctx: *mut tc_context, xyz: *const f32, n: usize, voxel_size: f32, out: *mut f32, n_out: *mut usize) -> c_int;
    pub fn tc_frame_stream_create(ctx: *mut tc_context, cfg: *const tc_frame_stream_config, out: *mut *mut tc_frame_stream) -> c_int;
    pub fn tc_frame_stream_send(s: *mut tc_frame_stream, frame: *const f32, n: usize, stride_floats: usize) -> c_int;
    pub fn tc_frame_stream_try_send(s: *mut tc_frame_stream, frame: *const f32, n: usize, stride_floats: usize, accepted: *mut c_int) -> c_int;
    pub fn tc_frame_stream_finish(s: *mut tc_frame_stream, results: *mut tc_frame_result, capacity: usize, n_results: *mut usize,
                                  metrics: *mut tc_frame_stream_metrics) -> c_int;
    pub fn tc_frame_stream_destroy(s: *mut tc_frame_stream);
    pub fn tc_read_kitti_bin(path: *const c_char, out_xyz: *mut f32, capacity_points: usize, n_points: *mut usize) -> c_int;
    // ---- stream ordering, communicator, one registration over several GPUs ----
    pub fn tc_context_wait_stream(ctx: *mut tc_context, other_hip_stream: *mut c_void) -> c_int;
    pub fn tc_stream_wait_context(ctx: *mut tc_context, other_hip_stream: *mut c_void) -> c_int;
    pub fn tc_comm_unique_id(id: *mut u8) -> c_int;
    pub fn tc_comm_create(ctx: *mut tc_context, nranks: c_int, rank: c_int, id: *const u8, out: *mut *mut tc_comm) -> c_int;
    pub fn tc_comm_adopt(ctx: *mut tc_context, nccl_comm: *mut c_void, nranks: c_int, rank: c_int, out: *mut *mut tc_comm) -> c_int;
    pub fn tc_comm_create_host(ctx: *mut tc_context, nranks: c_int, rank: c_int, f: tc_host_collective_fn, user: *mut c_void,
                               out: *mut *mut tc_comm) -> c_int;
    pub fn tc_comm_create_local(ctx: *mut tc_context, out: *mut *mut tc_comm) -> c_int;
    pub fn tc_comm_rank(comm: *const tc_comm) -> c_int;
    pub fn tc_comm_size(comm: *const tc_comm) -> c_int;
    pub fn tc_comm_destroy(comm: *mut tc_comm);
    pub fn tc_sharded_icp_point_to_plane_device(ctx: *mut tc_context, comm: *mut tc_comm, shard_mode: c_int, d_src: *const f32, ns: usize,
                                                d_tgt: *const f32, nt: usize, d_normals: *const f32, n_normals: usize, stride: usize,
                                                init: *const f32, max_iters: usize, max_dist: f32, conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_sharded_icp_detailed_device(ctx: *mut tc_context, comm: *mut tc_comm, shard_mode: c_int, d_src: *const f32, ns: usize,
                                          d_tgt: *const f32, nt: usize, init: *const f32, max_iters: usize, max_dist: f32, conv_thr: f32,
                                          res: *mut tc_icp_result) -> c_int;
    pub fn tc_sharded_estimate_normals_device(ctx: *mut tc_context, comm: *mut tc_comm, d_xyz: *const f32, n: usize,
                                              cfg: *const tc_normal_config, d_out: *mut f32) -> c_int;
    pub fn tc_sharded_estimate_normals_local_device(ctx: *mut tc_context, comm: *mut tc_comm, d_xyz: *const f32, n: usize,
                                                    cfg: *const tc_normal_config, d_out_slice: *mut f32, d_orig_index: *mut u32,
                                                    first: *mut usize, count: *mut usize) -> c_int;
    pub fn tc_debug_counter(ctx: *const tc_context, which: c_int) -> u64;
    // ---- device-resident cloud handles ----
    pub fn tc_cloud_upload(ctx: *mut tc_context, xyz: *const f32, n: usize, out: *mut *mut tc_cloud) -> c_int;
    pub fn tc_cloud_upload_device(ctx: *mut tc_context, d_xyz: *const f32, n: usize, out: *mut *mut tc_cloud) -> c_int;
    pub fn tc_cloud_size(cloud: *const tc_cloud) -> usize;
    pub fn tc_cloud_points_device(cloud: *const tc_cloud) -> *const f32;
    pub fn tc_cloud_normals_device(cloud: *const tc_cloud) -> *const f32;
    pub fn tc_cloud_estimate_normals(cloud: *mut tc_cloud, cfg: *const tc_normal_config, out: *mut f32) -> c_int;
    pub fn tc_cloud_estimate_normals_device(cloud: *mut tc_cloud, cfg: *const tc_normal_config, d_out: *mut f32) -> c_int;
    pub fn tc_cloud_set_normals_device(cloud: *mut tc_cloud, d_normals: *const f32, n_normals: usize, stride: usize) -> c_int;
    pub fn tc_cloud_icp_point_to_plane(source: *mut tc_cloud, target: *mut tc_cloud, init: *const f32, max_iters: usize, max_dist: f32,
                                       conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_cloud_icp_detailed(source: *mut tc_cloud, target: *mut tc_cloud, init: *const f32, max_iters: usize, max_dist: f32,
                                 conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_cloud_sharded_icp(comm: *mut tc_comm, shard_mode: c_int, point_to_plane: c_int, d_source: *const f32, n_source: usize,
                                target: *mut tc_cloud, init: *const f32, max_iters: usize, max_dist: f32, conv_thr: f32,
                                res: *mut tc_icp_result) -> c_int;
    pub fn tc_cloud_destroy(cloud: *mut tc_cloud);
    // ---- the rest of the header (device-pointer variants, batch, shard sessions, profiling): every tc_* export is declared ----
    pub fn tc_synchronize(ctx: *mut tc_context) -> c_int;
    pub fn tc_estimate_normals_device(ctx: *mut tc_context, d_xyz: *const f32, n: usize, cfg: *const tc_normal_config, d_out: *mut f32) -> c_int;
    pub fn tc_icp_detailed_device(ctx: *mut tc_context, d_src: *const f32, ns: usize, d_tgt: *const f32, nt: usize, init: *const f32,
                                  max_iters: usize, max_dist: f32, conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_icp_point_to_plane_detailed_device(ctx: *mut tc_context, d_src: *const f32, ns: usize, d_tgt: *const f32, nt: usize,
                                                 d_normals: *const f32, n_normals: usize, stride: usize, init: *const f32, max_iters: usize,
                                                 max_dist: f32, conv_thr: f32, res: *mut tc_icp_result) -> c_int;
    pub fn tc_batch_icp(ctxs: *const *mut tc_context, n_ctx: usize, jobs: *const tc_batch_icp_job, n_jobs: usize,
                        results: *mut tc_batch_icp_result) -> c_int;
    pub fn tc_icp_shard_create(ctx: *mut tc_context, point_to_plane: c_int, d_src_slice: *const f32, ns: usize, d_tgt: *const f32, nt: usize,
                               d_normals: *const f32, stride: usize, init: *const f32, max_dist: f32, conv_thr: f32,
                               out: *mut *mut tc_icp_shard) -> c_int;
    pub fn tc_icp_shard_sums(s: *mut tc_icp_shard) -> *mut f64;
    pub fn tc_icp_shard_reduce(s: *mut tc_icp_shard) -> c_int;
    pub fn tc_icp_shard_get_sums(s: *mut tc_icp_shard, d_out: *mut f64) -> c_int;
    pub fn tc_icp_shard_set_sums(s: *mut tc_icp_shard, d_in: *const f64) -> c_int;
    pub fn tc_icp_shard_done(s: *mut tc_icp_shard, done: *mut c_int) -> c_int;
    pub fn tc_icp_shard_apply(s: *mut tc_icp_shard) -> c_int;
    pub fn tc_icp_shard_finish(s: *mut tc_icp_shard, max_iters: usize, res: *mut tc_icp_result) -> c_int;
    pub fn tc_icp_shard_destroy(s: *mut tc_icp_shard);
    pub fn tc_gicp_device(ctx: *mut tc_context, d_src: *const f32, ns: usize, d_tgt: *const f32, nt: usize, init: *const f32,
                          cfg: *const tc_gicp_config, res: *mut tc_icp_result) -> c_int;
    pub fn tc_kiss_icp_device(ctx: *mut tc_context, d_src: *const f32, ns: usize, d_tgt: *const f32, nt: usize, init: *const f32,
                              cfg: *const tc_kiss_icp_config, res: *mut tc_icp_result, n_source_down: *mut usize) -> c_int;
    pub fn tc_knn_device(ctx: *mut tc_context, d_cloud: *const f32, n: usize, d_queries: *const f32, nq: usize, k: usize,
                         d_idx: *mut u32, d_dist: *mut f32, d_count: *mut u32) -> c_int;
    pub fn tc_radius_search_device(ctx: *mut tc_context, d_cloud: *const f32, n: usize, d_queries: *const f32, nq: usize, radius: f32, k_max: usize,
                                   d_idx: *mut u32, d_dist: *mut f32, d_count: *mut u32) -> c_int;
    pub fn tc_search_index_create_device(ctx: *mut tc_context, d_cloud: *const f32, n: usize, k_hint: usize, out: *mut *mut tc_search_index) -> c_int;
    pub fn tc_search_index_query_device(index: *mut tc_search_index, d_queries: *const f32, nq: usize, k: usize, radius: f32,
                                        d_idx: *mut u32, d_dist: *mut f32, d_count: *mut u32) -> c_int;
    pub fn tc_voxel_grid_filter_device(ctx: *mut tc_context, d_xyz: *const f32, n: usize, voxel_size: f32, d_out: *mut f32, n_out: *mut usize) -> c_int;
    pub fn tc_profile_enable(ctx: *mut tc_context, on: c_int);
    pub fn tc_profile_reset(ctx: *mut tc_context);
    pub fn tc_profile_read(ctx: *mut tc_context, out: *mut tc_kernel_stat, capacity: usize) -> usize;
}
