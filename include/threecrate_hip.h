/*
 * threecrate_hip.h -- C ABI of the MI355X (gfx950) normals + ICP backend for threecrate.
 *
 * This is the drop-in boundary.  The reference has no FFI on this
 * path; callers link Rust crates and call free functions / GpuContext methods.  Each entry
 * point below names the reference signature it stands in for (paths relative to
 * /root/reference).  A Rust `extern "C"` shim maps them 1:1 back onto
 *   threecrate-algorithms: estimate_normals / estimate_normals_with_config /
 *                          estimate_normals_radius / icp / icp_detailed /
 *                          icp_point_to_point / icp_point_to_plane(_detailed)
 *   threecrate-gpu:        GpuContext::new, gpu_estimate_normals, gpu_icp,
 *                          gpu_icp_point_to_plane, gpu_batch_icp
 * plus what the reference does not have and an MI355X node needs: device-resident cloud handles (tc_cloud_*: one index build
 * per cloud), a communicator (tc_comm_*: RCCL bound at run time) and ONE registration / ONE cloud's normals over the GPUs of a
 * node (tc_sharded_*, tc_cloud_sharded_icp) with the per-iteration all-reduce inside the library.
 *
 * Memory layouts (threecrate-core):
 *   Point3f         = 3 x f32, AoS               (threecrate-core/src/point.rs:8)
 *   NormalPoint3f   = {position[3], normal[3]}   (threecrate-core/src/point.rs:31-36, #[repr(C)])
 *   Isometry3<f32>  = unit quaternion (i, j, k, w) + translation (x, y, z) = 7 x f32
 *                     (nalgebra storage order; re-export threecrate-core/src/lib.rs:26)
 *
 * All calls are blocking (like the reference's `device.poll(Wait)`), never throw or abort
 * across the ABI, and report errors as tc_status + tc_last_error_message().  A tc_context is
 * bound to one HIP device and one stream and is NOT thread-safe (one context per thread /
 * GPU, like one GpuContext); different contexts may be used concurrently.
 *
 * There is NO CPU fallback: without a HIP device every compute entry point returns TC_GPU.
 */
#ifndef THREECRATE_HIP_H
#define THREECRATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_ABI_VERSION 2

/* threecrate_core::Error variants used on the path (threecrate-core/src/error.rs:7-28) */
typedef enum tc_status {
    TC_OK = 0,
    TC_INVALID_DATA = 1,   /* Error::InvalidData  */
    TC_ALGORITHM = 2,      /* Error::Algorithm    */
    TC_GPU = 3,            /* Error::Gpu (HIP / device failure) */
    TC_UNSUPPORTED = 4     /* Error::Unsupported  */
} tc_status;

typedef struct tc_context tc_context;

/* NormalEstimationConfig (threecrate-algorithms/src/normals.rs:17-37); defaults via tc_normal_config_default */
typedef struct tc_normal_config {
    uint64_t k_neighbors;             /* default 10 */
    float    radius;                  /* Option<f32>: valid iff has_radius */
    int32_t  has_radius;
    int32_t  consistent_orientation;  /* default 1 */
    int32_t  has_viewpoint;           /* Option<Point3f>: valid iff has_viewpoint */
    float    viewpoint[3];
} tc_normal_config;

/* ICPResult (threecrate-algorithms/src/registration.rs:13-24) */
typedef struct tc_icp_result {
    float    transformation[7];  /* Isometry3<f32>: qi qj qk qw tx ty tz */
    float    mse;
    uint64_t iterations;
    int32_t  converged;
    uint64_t n_correspondences;  /* number of valid pairs of the last executed iteration */
    /* Optional, caller-allocated, n_source entries (HOST memory for the host entry points,
       DEVICE memory for the *_device entry points): matched target index per source point,
       0xFFFFFFFF = no correspondence.  The reference's Vec<(usize,usize)> is the list of
       (j, corr_target[j]) with corr_target[j] != 0xFFFFFFFF in ascending j. */
    uint32_t *corr_target;
} tc_icp_result;

/* BatchICPJob / BatchICPResult (threecrate-gpu/src/icp.rs:132-147) */
typedef struct tc_batch_icp_job {
    const float *source; size_t n_source;
    const float *target; size_t n_target;
    size_t max_iterations;
    float  convergence_threshold;
    float  max_correspondence_distance;   /* < 0 : none */
} tc_batch_icp_job;
typedef struct tc_batch_icp_result {
    float    transformation[7];
    float    final_error;
    uint64_t iterations;
    int32_t  status;                      /* tc_status of this job */
} tc_batch_icp_result;

/* per-kernel timing (hipEvent on the context's stream), for roofline reporting */
typedef struct tc_kernel_stat {
    char     name[48];
    uint64_t launches;
    double   total_ms;
    double   min_ms, max_ms;     /* shortest / longest timed launch (ABI version 2) */
} tc_kernel_stat;

/* ---- context: GpuContext::new (threecrate-gpu/src/device.rs:16-50) ---- */
int         tc_abi_version(void);
int         tc_device_count(void);
tc_status   tc_context_create(int device, tc_context **out);
/* same, but run on a caller-owned hipStream_t (e.g. torch's current stream) */
tc_status   tc_context_create_on_stream(int device, void *hip_stream, tc_context **out);
/* Stream ordering rule for the *_device entry points: the library reads the caller's device buffers on the CONTEXT's
 * stream.  Work that produces those buffers on another stream (a torch op on torch's current stream, a hipMemcpyAsync on
 * the caller's own stream) must be ordered before the call: either create the context on that stream
 * (tc_context_create_on_stream), synchronise the producer stream, or call tc_context_wait_stream(ctx, producer_stream):
 * it records an event on `other_stream` and makes the context's stream wait for it (no host wait).  Every entry point
 * returns with its outputs complete (the context's stream has been synchronised), so nothing is needed on the way out. */
tc_status   tc_context_wait_stream(tc_context *ctx, void *other_hip_stream);
/* The other direction: whatever is enqueued on `other_hip_stream` after this call waits for the work the context's stream holds now.
 * For a producer that is about to overwrite or free a device buffer it has just handed to a stream-ordered entry point
 * (tc_cloud_upload_device copies on the context's stream): no host wait. */
tc_status   tc_stream_wait_context(tc_context *ctx, void *other_hip_stream);
/* Device memory given back by destroyed handles (tc_cloud, tc_search_index) is parked in the context and reused by the next
 * handle of a similar size (a handle per frame costs no hipMalloc); the parked amount is capped at a few handles' worth.
 * tc_context_trim releases all of it to the device -- for a caller about to hand the GPU to another allocator. */
tc_status   tc_context_trim(tc_context *ctx);
void        tc_context_destroy(tc_context *ctx);
const char *tc_last_error_message(const tc_context *ctx);
tc_status   tc_synchronize(tc_context *ctx);
void        tc_normal_config_default(tc_normal_config *cfg);   /* normals.rs:28-36 */

/* ---- normals ----
 * estimate_normals_with_config(&PointCloud<Point3f>, &NormalEstimationConfig)
 *     -> Result<PointCloud<NormalPoint3f>>        (normals.rs:257-357)
 * estimate_normals (normals.rs:238-247) = cfg{k}, estimate_normals_radius (normals.rs:368-380)
 * = cfg{k=10, radius, consistent, viewpoint None}; gpu_estimate_normals
 * (threecrate-gpu/src/normals.rs:443-461) has the same meaning.
 * xyz: n x 3 f32.  out: n x 6 f32 (NormalPoint3f).  n == 0 -> TC_OK before the k check.
 * Limits of this backend (the reference has none): k_neighbors <= 2047 (up to 128 the k + 1 nearest incl. the point itself live in
 * a 129-entry register list, one lane per point; beyond that a block-per-point kernel serves every point; the k-NN / radius
 * exports below return up to 2048 entries per query) -> TC_UNSUPPORTED (an error with a message, never undefined behaviour) beyond.
 * k_neighbors > 128 together with a radius: the block-per-point kernel folds the radius ball itself (no cap on its members).
 * Non-finite points (NaN / +-inf coordinates) are inert: never a neighbour, their own normal is the default (0, 0, 1). */
tc_status tc_estimate_normals(tc_context *ctx, const float *xyz, size_t n,
                              const tc_normal_config *cfg, float *out_normal_points);
/* same with xyz / out already resident in device memory (HBM) */
tc_status tc_estimate_normals_device(tc_context *ctx, const float *d_xyz, size_t n,
                                     const tc_normal_config *cfg, float *d_out_normal_points);

/* Normals of ONE big cloud over several GPUs (the reference has no multi-device code): the cloud is
 * replicated, every rank builds the same index (the cell-sorted order is deterministic) and computes the normals of the
 * cell-sorted positions [begin, end) into d_slice_out ((end - begin) x 6 floats: position, normal -- in SORTED order);
 * after an all-gather of the slices (RCCL; sizes are known: shard r = [r n / W, (r + 1) n / W)) every rank calls
 * tc_normals_unsort_device to obtain the NormalPoint3f array in input order.  begin == 0, end == n on one GPU followed by
 * the unsort is tc_estimate_normals_device.  The unsort uses the index the slice call left in the context. */
tc_status tc_estimate_normals_slice_device(tc_context *ctx, const float *d_xyz, size_t n, const tc_normal_config *config,
                                           size_t begin, size_t end, float *d_slice_out);
tc_status tc_normals_unsort_device(tc_context *ctx, const float *d_sorted_all, size_t n, float *d_out);

/* ---- ICP point-to-point ----
 * icp_detailed(source, target, init, max_iters, max_correspondence_distance: Option<f32>,
 *              convergence_threshold) -> Result<ICPResult>   (registration.rs:258-370)
 * max_correspondence_distance < 0 encodes None. */
tc_status tc_icp_detailed(tc_context *ctx, const float *source, size_t n_source,
                          const float *target, size_t n_target, const float init[7],
                          size_t max_iters, float max_correspondence_distance,
                          float convergence_threshold, tc_icp_result *result);
tc_status tc_icp_detailed_device(tc_context *ctx, const float *d_source, size_t n_source,
                          const float *d_target, size_t n_target, const float init[7],
                          size_t max_iters, float max_correspondence_distance,
                          float convergence_threshold, tc_icp_result *result);
/* icp_point_to_point(source, target, init, max_iterations, convergence_threshold,
 *                    max_correspondence_distance) (registration.rs:644-680):
 * adds the convergence_threshold <= 0 -> InvalidData check.  gpu_icp
 * (threecrate-gpu/src/icp.rs:977-994) maps here with init = identity. */
tc_status tc_icp_point_to_point(tc_context *ctx, const float *source, size_t n_source,
                          const float *target, size_t n_target, const float init[7],
                          size_t max_iterations, float convergence_threshold,
                          float max_correspondence_distance, tc_icp_result *result);
/* icp(source, target, init, max_iters) -> Isometry3 (registration.rs:232-242):
 * threshold 1e-6, no cut-off, any error returns `init`.  Always TC_OK unless ctx is NULL. */
tc_status tc_icp(tc_context *ctx, const float *source, size_t n_source,
                 const float *target, size_t n_target, const float init[7],
                 size_t max_iters, float out_transformation[7]);

/* ---- ICP point-to-plane ----
 * icp_point_to_plane_detailed(source, target, target_normals: &[Vector3f], init, max_iters,
 *     max_correspondence_distance, convergence_threshold)    (registration.rs:508-602)
 * icp_point_to_plane (registration.rs:488-496) = (.., None -> -1.0f, 1e-6f);
 * gpu_icp_point_to_plane (threecrate-gpu/src/icp.rs:1017-1036) has the same meaning.
 * n_target_normals != n_target -> TC_INVALID_DATA (checked before max_iters).
 * normal_stride = floats between consecutive normals: 3 for &[Vector3f]; 6 lets a caller pass
 * &NormalPoint3f[0].normal of estimate_normals' output directly. */
tc_status tc_icp_point_to_plane_detailed(tc_context *ctx, const float *source, size_t n_source,
                          const float *target, size_t n_target,
                          const float *target_normals, size_t n_target_normals, size_t normal_stride,
                          const float init[7], size_t max_iters,
                          float max_correspondence_distance, float convergence_threshold,
                          tc_icp_result *result);
tc_status tc_icp_point_to_plane_detailed_device(tc_context *ctx, const float *d_source, size_t n_source,
                          const float *d_target, size_t n_target,
                          const float *d_target_normals, size_t n_target_normals, size_t normal_stride,
                          const float init[7], size_t max_iters,
                          float max_correspondence_distance, float convergence_threshold,
                          tc_icp_result *result);

/* ---- device-resident cloud handles ("tc_cloud_upload -> handle, so frames can stay on the GPU") ----
 * The reference passes &PointCloud<Point3f> and rebuilds its KdTree inside every call (normals.rs:272, registration.rs:281,
 * :536).  A tc_cloud owns a device copy of the points and is indexed ONCE: tc_cloud_estimate_normals leaves the cell-sorted
 * records AND the cell-sorted normals in the handle -- the layout the ICP kernels read -- so the registration of the next frame
 * against it needs no second index build and no normals gather.  Same answers as the handle-free entry points: the cell edge
 * of the shared grid changes speed, not results -- except where two candidates are at EXACTLY the same distance and lie in different
 * cells (the lower position in the cell-sorted order wins, and that order belongs to the grid; the reference's own choice there is
 * its heap's): a handful of rows per million points (tools/dev/paths_stress.py).
 *   tc_cloud_estimate_normals[_device](c, cfg, out): estimate_normals_with_config (normals.rs:257-357); out (n x 6
 *       NormalPoint3f, host / device) may be NULL when only the handle needs them (the 24-byte scattered stores are skipped).
 *   tc_cloud_set_normals_device: normals computed elsewhere (n x 3 with stride 3, or &NormalPoint3f[0].normal with stride 6).
 *   tc_cloud_icp_point_to_plane(source, target, ...): icp_point_to_plane_detailed (registration.rs:508-602) with the target's
 *       normals; a target without normals -> TC_INVALID_DATA (the reference's length check, :522-526).
 *   tc_cloud_icp_detailed(source, target, ...): icp_detailed (registration.rs:258-370).
 *   result->corr_target, when given, is DEVICE memory (n_source entries).  max_correspondence_distance < 0: None.
 * Handles belong to their context (same thread rule); destroy them before the context. */
typedef struct tc_cloud tc_cloud;
tc_status    tc_cloud_upload(tc_context *ctx, const float *xyz, size_t n, tc_cloud **out);            /* host -> HBM */
tc_status    tc_cloud_upload_device(tc_context *ctx, const float *d_xyz, size_t n, tc_cloud **out);   /* HBM -> HBM copy */
size_t       tc_cloud_size(const tc_cloud *cloud);
const float *tc_cloud_points_device(const tc_cloud *cloud);      /* n x 3, device */
const float *tc_cloud_normals_device(const tc_cloud *cloud);     /* n x 6 NormalPoint3f in input order; NULL if the handle has no normals.
                                                                  * Only the host-output estimate keeps that copy; otherwise it is made from the
                                                                  * cell-sorted normals by this call (one kernel + a stream synchronisation):
                                                                  * despite the const handle the call then allocates, launches and waits (same
                                                                  * thread rule as every other call on the context), and NULL means failure with
                                                                  * the reason in tc_last_error_message; "no normals" leaves the message alone. */
tc_status    tc_cloud_estimate_normals(tc_cloud *cloud, const tc_normal_config *config, float *out_normal_points);
tc_status    tc_cloud_estimate_normals_device(tc_cloud *cloud, const tc_normal_config *config, float *d_out_normal_points);
tc_status    tc_cloud_set_normals_device(tc_cloud *cloud, const float *d_normals, size_t n_normals, size_t normal_stride);
tc_status    tc_cloud_icp_point_to_plane(tc_cloud *source, tc_cloud *target, const float init[7], size_t max_iters,
                                         float max_correspondence_distance, float convergence_threshold, tc_icp_result *result);
tc_status    tc_cloud_icp_detailed(tc_cloud *source, tc_cloud *target, const float init[7], size_t max_iters,
                                   float max_correspondence_distance, float convergence_threshold, tc_icp_result *result);
void         tc_cloud_destroy(tc_cloud *cloud);

/* gpu_batch_icp(&GpuContext, &[BatchICPJob]) (threecrate-gpu/src/icp.rs:997-1002, 151-185):
 * job i runs on ctxs[i % n_ctx]; jobs on different contexts (GPUs) run concurrently. */
tc_status tc_batch_icp(tc_context *const *ctxs, size_t n_ctx, const tc_batch_icp_job *jobs,
                       size_t n_jobs, tc_batch_icp_result *results);

/* ---- sharded ICP building blocks (one big cloud over several GPUs) ----
 * A shard session holds the replicated target index and this rank's slice of source points.
 * Per iteration the host (threecrate_amd.distributed) calls reduce -> all-reduce of the
 * packed sums (29 f64 words p2plane / 17 p2p) over RCCL -> apply.  Every rank applies the
 * identical reduced buffer, so all ranks hold the same transform without a broadcast. */
typedef struct tc_icp_shard tc_icp_shard;
#define TC_ICP_SUMS_P2PLANE 29
#define TC_ICP_SUMS_P2P     17
#define TC_ICP_SUMS_STRIDE  32
tc_status tc_icp_shard_create(tc_context *ctx, int point_to_plane,
                              const float *d_source_slice, size_t n_source_slice,
                              const float *d_target, size_t n_target,
                              const float *d_target_normals, size_t normal_stride,
                              const float init[7], float max_correspondence_distance,
                              float convergence_threshold, tc_icp_shard **out);
/* device pointer to TC_ICP_SUMS_STRIDE doubles: this rank's packed sums (all-reduce in place) */
double     *tc_icp_shard_sums(tc_icp_shard *s);
tc_status   tc_icp_shard_reduce(tc_icp_shard *s);              /* correspondences + local sums */
/* async device-to-device copies of the TC_ICP_SUMS_STRIDE packed sums on the context's stream,
   for callers that all-reduce in their own buffer (e.g. a torch tensor) */
tc_status   tc_icp_shard_get_sums(tc_icp_shard *s, double *d_out);
tc_status   tc_icp_shard_set_sums(tc_icp_shard *s, const double *d_in);
/* blocking: *done = 1 once the loop has converged or failed (identical on every rank) */
tc_status   tc_icp_shard_done(tc_icp_shard *s, int *done);
tc_status   tc_icp_shard_apply(tc_icp_shard *s);               /* solve + compose + convergence (device) */
tc_status   tc_icp_shard_finish(tc_icp_shard *s, size_t max_iters, tc_icp_result *result);
void        tc_icp_shard_destroy(tc_icp_shard *s);

/* ---- communicator: one rank per GPU of one node (north_star: "a single RCCL all-reduce of the 6x6 system
 * per ICP iteration over xGMI") ----
 * The reference has no multi-device code (threecrate-gpu/src/icp.rs:151-185 runs its batch jobs one after the other on
 * one device), so there is no reference signature to mirror: this is the boundary a Rust host (one process or thread per
 * GPU) binds to run ONE registration over several GPUs.  A tc_comm wraps an RCCL communicator bound to the context's
 * device; its collectives are enqueued on the context's stream (no host synchronisation inside the ICP loop).
 *   rank 0:  tc_comm_unique_id(id)  -> the host distributes the 128 bytes by any means (pipe, file, MPI, torch.distributed)
 *   all:     tc_comm_create(ctx, nranks, rank, id, &comm)          = ncclCommInitRank
 *   or:      tc_comm_adopt(ctx, existing ncclComm_t, nranks, rank, &comm)   (the host already owns a communicator)
 *   or:      tc_comm_create_host(...): the collectives go through a HOST callback on pinned buffers (hosts whose ranks are
 *            connected by something else than RCCL -- MPI, gloo, shared memory; several ranks on one GPU in tests).
 *            One blocking device<->host round trip per collective: correct, not fast.
 *   or:      tc_comm_create_local(ctx, &comm): nranks = 1, every collective is a no-op.
 * librccl is resolved at run time (symbols already in the process first, then librccl.so.1): the library loads and every
 * single-GPU entry point works on a machine without RCCL; tc_comm_unique_id / tc_comm_create then return TC_UNSUPPORTED. */
typedef struct tc_comm tc_comm;
#define TC_COMM_ID_BYTES 128
typedef enum tc_coll_op {
    TC_COLL_SUM_F64 = 0,        /* buf: count doubles, summed over the ranks in place */
    TC_COLL_SUM_U32 = 1,        /* buf: count uint32, summed in place */
    TC_COLL_ALLGATHER_U8 = 2    /* buf: nranks x count bytes; rank r's part sits at r * count; in place */
} tc_coll_op;
/* returns 0 on success; called from the thread that called the sharded entry point */
typedef int (*tc_host_collective_fn)(void *user, int op, void *host_buf, size_t count);
tc_status tc_comm_unique_id(uint8_t id[TC_COMM_ID_BYTES]);
tc_status tc_comm_create(tc_context *ctx, int nranks, int rank, const uint8_t id[TC_COMM_ID_BYTES], tc_comm **out);
tc_status tc_comm_adopt(tc_context *ctx, void *nccl_comm, int nranks, int rank, tc_comm **out);
tc_status tc_comm_create_host(tc_context *ctx, int nranks, int rank, tc_host_collective_fn fn, void *user, tc_comm **out);
tc_status tc_comm_create_local(tc_context *ctx, tc_comm **out);
int       tc_comm_rank(const tc_comm *comm);
int       tc_comm_size(const tc_comm *comm);
void      tc_comm_destroy(tc_comm *comm);

/* ---- ONE registration over all ranks of a communicator (BASELINE configs[3]: a 10 M-point cloud over 8 GPUs) ----
 * icp_point_to_plane_detailed (registration.rs:508-602) / icp_detailed (:258-370) with the SOURCE points sharded over
 * the ranks and the target (+ normals + grid) replicated.  Per iteration every rank reduces its shard to the packed
 * normal equations (TC_ICP_SUMS_P2PLANE / _P2P f64 words), ONE all-reduce(sum) of TC_ICP_SUMS_STRIDE doubles on the
 * compute stream makes them global, and every rank applies the identical buffer (same solve, same convergence test: no
 * broadcast).  The host polls the `done` flag two chunks of iterations behind, like the single-GPU loop: no host wait
 * sits between an iteration's kernels and its collective.  Every rank returns the same transformation / mse /
 * iterations / converged / n_correspondences.
 *   TC_SHARD_SPATIAL: every rank passes the SAME full source cloud; rank r takes the contiguous range
 *       [r ns / W, (r + 1) ns / W) of the source ordered by the target cell of its initially transformed position (tile
 *       major): a spatially compact shard.  corr_target (device, n_source entries, optional) is complete on every rank
 *       (one all-reduce of n_source words after the loop).
 *   TC_SHARD_LOCAL: every rank passes ITS OWN part of the source (any partition; may be empty on some ranks);
 *       corr_target then has n_source (local) entries.
 *   TC_SHARD_INDEX: every rank passes the SAME full source cloud; rank r takes the ORIGINAL-index range
 *       [r rows, min((r + 1) rows, ns)), rows = ceil(ns / W), and orders only those points by target cell: the per-call set-up
 *       (the source's counting sort) shrinks with 1 / W like the iterations do -- with TC_SHARD_SPATIAL every rank sorts the
 *       whole source.  corr_target (device, n_source entries, optional) is completed on every rank by ONE all-gather of the
 *       ranks' slices (4 ns bytes in total) instead of an all-reduce of n_source words.  The default of the Python mirror for
 *       more than one rank.
 * Point-to-point: the post-loop mse of a run that did not converge (registration.rs:343-361) is reduced over the ranks
 * too.  All ranks must call with the same arguments apart from the source in TC_SHARD_LOCAL mode; validation failures are
 * the single-GPU entry points' and identical on every rank. */
typedef enum tc_shard_mode { TC_SHARD_SPATIAL = 0, TC_SHARD_LOCAL = 1, TC_SHARD_INDEX = 2 } tc_shard_mode;
tc_status tc_sharded_icp_point_to_plane_device(tc_context *ctx, tc_comm *comm, int shard_mode,
                          const float *d_source, size_t n_source, const float *d_target, size_t n_target,
                          const float *d_target_normals, size_t n_target_normals, size_t normal_stride,
                          const float init[7], size_t max_iters, float max_correspondence_distance,
                          float convergence_threshold, tc_icp_result *result);
tc_status tc_sharded_icp_detailed_device(tc_context *ctx, tc_comm *comm, int shard_mode,
                          const float *d_source, size_t n_source, const float *d_target, size_t n_target,
                          const float init[7], size_t max_iters, float max_correspondence_distance,
                          float convergence_threshold, tc_icp_result *result);
/* estimate_normals_with_config (normals.rs:257-357) of one replicated cloud over the ranks: every rank builds the same
 * index, computes the records of ITS range of cell-sorted positions, one all-gather (n x 24 bytes in total) on the compute
 * stream, a local kernel restores the input order.  Every rank receives all n records in d_out (n x 6). */
tc_status tc_sharded_estimate_normals_device(tc_context *ctx, tc_comm *comm, const float *d_xyz, size_t n,
                                             const tc_normal_config *config, float *d_out_normal_points);
/* The same without the all-gather (240 MB at 10 M points: it, not the kernel, bounds the call on 8 GPUs): this rank's records only.
 * d_out_slice: ceil(n / W) x 6 floats (position, normal) for the cell-sorted positions [*first, *first + *count);
 * d_orig_index (optional): the input index of each of those records.  No collective is issued. */
tc_status tc_sharded_estimate_normals_local_device(tc_context *ctx, tc_comm *comm, const float *d_xyz, size_t n,
                                                   const tc_normal_config *config, float *d_out_slice, uint32_t *d_orig_index,
                                                   size_t *first, size_t *count);

/* Diagnostics (tests): work counters of a context since its creation.  TC_COUNTER_INDEXED_POINTS: points that went through an
 * index build (counting sort) -- the per-call set-up work of the sharded entry points, which must shrink with the rank count. */
typedef enum tc_counter {
    TC_COUNTER_INDEXED_POINTS = 0, TC_COUNTER_INDEX_BUILDS = 1,
    /* Search statistics of the ICP main pass (SURVEY 8d's secondary figures), summed over the registrations run while the context
     * is in profiling mode 3 (tc_profile_enable(ctx, 3) zeroes them and selects the kernel's counting instantiation -- a few per
     * cent slower, same results; any other mode runs the product's kernel, which carries no counter):
     * iterations executed | wave trips (64 source points each) | trips in which no lane searched (every previous match was kept) |
     * searches (lanes that scanned the grid) | candidate steps those searches needed (4 distance evaluations each) | candidate
     * steps the trips took = the steps of each trip's slowest lane (x 64 / steps needed = the lock-step ratio). */
    TC_COUNTER_ICP_ITERATIONS = 2, TC_COUNTER_ICP_TRIPS = 3, TC_COUNTER_ICP_TRIPS_WITHOUT_SEARCH = 4, TC_COUNTER_ICP_SEARCHES = 5,
    TC_COUNTER_ICP_STEPS_NEEDED = 6, TC_COUNTER_ICP_STEPS_TAKEN = 7
} tc_counter;
unsigned long long tc_debug_counter(const tc_context *ctx, int which);

/* One registration over the ranks of a communicator (see tc_sharded_icp_point_to_plane_device) against a TARGET HANDLE: every
 * rank holds the same target cloud in its own handle, whose index, cell-sorted normals and inscribed-ball bounds are built once
 * per handle instead of once per call (a map many scans are registered against).  point_to_plane != 0 needs normals in the handle.
 * The source is a device buffer, sharded per shard_mode; corr_target (device, optional) as in the buffer-based entry points. */
tc_status    tc_cloud_sharded_icp(tc_comm *comm, int shard_mode, int point_to_plane, const float *d_source, size_t n_source,
                                  tc_cloud *target, const float init[7], size_t max_iters, float max_correspondence_distance,
                                  float convergence_threshold, tc_icp_result *result);

/* ---- multiscale ICP ----
 * multiscale_icp_point_to_point(source, target, init, &MultiScaleIcpConfig) -> Result<ICPResult>
 * (threecrate-algorithms/src/registration.rs:704-789; config :26-71): per level voxel_grid_filter both
 * clouds, icp_point_to_point from the running transform, then a full-resolution refinement;
 * iterations are summed.  max_correspondence_distance < 0 encodes None. */
typedef struct tc_icp_scale_level {          /* IcpScaleLevel, registration.rs:27-35 */
    float  voxel_size;
    size_t max_iterations;
    float  max_correspondence_distance;
} tc_icp_scale_level;
typedef struct tc_multiscale_icp_config {    /* MultiScaleIcpConfig, registration.rs:38-44 */
    const tc_icp_scale_level *levels;
    size_t n_levels;
    size_t final_refinement_iterations;
    float  final_max_correspondence_distance;
    float  convergence_threshold;
} tc_multiscale_icp_config;
tc_status tc_multiscale_icp_point_to_point(tc_context *ctx, const float *source, size_t n_source,
                          const float *target, size_t n_target, const float init[7],
                          const tc_multiscale_icp_config *config, tc_icp_result *result);

/* ---- GICP ----
 * gicp(source, target, init, GicpConfig) -> Result<ICPResult>  (threecrate-algorithms/src/gicp.rs:100-305; config
 * :25-40, defaults 50 iterations, max distance 1.0, threshold 1e-6, k = 20): per-point covariances from the k
 * nearest points (:52-86), per pair M = C_t + R C_s R^T, 6x6 Gauss-Newton system H dx = g, Cholesky then LU,
 * update Rz Ry Rx + t, mse = mean squared correspondence distance before the update.  Clouds smaller than
 * max(k, 4) points or with a bounding-box side < 1e-4 -> TC_INVALID_DATA; k > 2048 -> TC_UNSUPPORTED. */
typedef struct tc_gicp_config {
    size_t max_iterations;
    float  max_correspondence_distance;
    float  convergence_threshold;
    size_t k_correspondences;
} tc_gicp_config;
tc_status tc_gicp(tc_context *ctx, const float *source, size_t n_source, const float *target, size_t n_target,
                  const float init[7], const tc_gicp_config *config, tc_icp_result *result);
tc_status tc_gicp_device(tc_context *ctx, const float *d_source, size_t n_source, const float *d_target, size_t n_target,
                         const float init[7], const tc_gicp_config *config, tc_icp_result *result);

/* ---- KISS-ICP ----
 * kiss_icp(source, target, init, KissIcpConfig) -> Result<ICPResult>  (threecrate-algorithms/src/kiss_icp.rs:
 * 183-300; config :28-49, defaults voxel 1.0, max_range 100, min_range 0.5, 50 iterations): range filter
 * (:56-70) and voxel down-sampling of the source, adaptive correspondence threshold from `init` (:82-95),
 * point-to-point updates against the full target, mse measured AFTER each update, |prev - mse| < 1e-6 rule,
 * not converged -> last mse.  correspondences index the DOWN-SAMPLED source (here: voxels in (kx, ky, kz)
 * order; the reference's HashMap order is unspecified): corr_target (capacity n_source) gets
 * *n_source_down entries. */
typedef struct tc_kiss_icp_config {
    float  voxel_size;
    float  max_range;
    float  min_range;
    size_t max_iterations;
} tc_kiss_icp_config;
tc_status tc_kiss_icp(tc_context *ctx, const float *source, size_t n_source, const float *target, size_t n_target,
                      const float init[7], const tc_kiss_icp_config *config, tc_icp_result *result, size_t *n_source_down);
tc_status tc_kiss_icp_device(tc_context *ctx, const float *d_source, size_t n_source, const float *d_target, size_t n_target,
                             const float init[7], const tc_kiss_icp_config *config, tc_icp_result *result, size_t *n_source_down);

/* ---- batch k-NN export ----
 * NearestNeighborSearch::find_k_nearest(&query, k) -> Vec<(usize, f32)> for many queries
 * (threecrate-core/src/traits.rs:6-12, threecrate-algorithms/src/nearest_neighbor.rs:177-251;
 * gpu_find_k_nearest_batch threecrate-gpu/src/nearest_neighbor.rs:345-355; Python KdTree.knn
 * threecrate-python/src/lib.rs:735-745).  Row q of idx / dist (nq x k) holds count[q] = min(k, n)
 * neighbours in ascending distance = sqrt(d2); k == 0 or n == 0 -> count = 0.  k <= 2048 (up to 129 in a lane-per-query
 * kernel, beyond that a block per query). */
tc_status tc_knn(tc_context *ctx, const float *cloud, size_t n, const float *queries, size_t nq, size_t k,
                 uint32_t *idx, float *dist, uint32_t *count);
tc_status tc_knn_device(tc_context *ctx, const float *d_cloud, size_t n, const float *d_queries, size_t nq, size_t k,
                        uint32_t *d_idx, float *d_dist, uint32_t *d_count);

/* ---- radius search export ----
 * NearestNeighborSearch::find_radius_neighbors(&query, radius) -> Vec<(usize, f32)> (traits.rs:6-12,
 * nearest_neighbor.rs:254-298: every point with d2 <= radius^2, ascending distance) for many queries, capped at
 * k_max nearest per query like gpu_find_radius_neighbors (threecrate-gpu/src/nearest_neighbor.rs:357-367,
 * k_max = 32 there).  Row q of idx / dist (nq x k_max) holds count[q] entries; count[q] == k_max may be a
 * truncated neighbourhood.  radius <= 0 -> empty results.  k_max <= 2048. */
tc_status tc_radius_search(tc_context *ctx, const float *cloud, size_t n, const float *queries, size_t nq, float radius, size_t k_max,
                           uint32_t *idx, float *dist, uint32_t *count);
tc_status tc_radius_search_device(tc_context *ctx, const float *d_cloud, size_t n, const float *d_queries, size_t nq, float radius,
                                  size_t k_max, uint32_t *d_idx, float *d_dist, uint32_t *d_count);

/* ---- persistent search index (the NearestNeighborSearch object) ----
 * KdTree::new(&points) once (threecrate-algorithms/src/nearest_neighbor.rs:37-58), then any number of
 * find_k_nearest / find_radius_neighbors calls (threecrate-core/src/traits.rs:6-12; Python `KdTree`
 * threecrate-python/src/lib.rs:707-776).  The handle owns a cell-sorted copy of the cloud and its grid in device
 * memory; the caller's buffer is not referenced after create.  k_hint sizes the cells (any k <= 2048 is answered
 * exactly whatever the hint).  An empty cloud gives an empty index (every count = 0), like the reference.
 * query: radius < 0 -> the k nearest; radius >= 0 -> the neighbours with distance <= radius among the k nearest
 * (count[q] == k means there may be more).  Rows of idx / dist (nq x k) ascending by distance = sqrt(d2).
 * A handle belongs to its context (same thread rule); destroy it before the context. */
typedef struct tc_search_index tc_search_index;
tc_status tc_search_index_create(tc_context *ctx, const float *cloud, size_t n, size_t k_hint, tc_search_index **out);
tc_status tc_search_index_create_device(tc_context *ctx, const float *d_cloud, size_t n, size_t k_hint, tc_search_index **out);
size_t tc_search_index_size(const tc_search_index *index);
tc_status tc_search_index_query(tc_search_index *index, const float *queries, size_t nq, size_t k, float radius,
                                uint32_t *idx, float *dist, uint32_t *count);
tc_status tc_search_index_query_device(tc_search_index *index, const float *d_queries, size_t nq, size_t k, float radius,
                                       uint32_t *d_idx, float *d_dist, uint32_t *d_count);
/* find_radius_neighbors without a cap (nearest_neighbor.rs:254-298: every point with d2 <= radius^2): count per query,
 * then fill -- the caller turns the counts into offsets (exclusive prefix sum, `total` = their sum) and receives
 * (original index, distance = sqrt(d2)) at [offsets[q], offsets[q] + counts[q]) in the index's scan order; sort a
 * query's segment by distance to obtain the reference's order.  Host buffers.  radius <= 0 -> all counts 0. */
tc_status tc_search_index_radius_count(tc_search_index *index, const float *queries, size_t nq, float radius, uint32_t *counts);
tc_status tc_search_index_radius_fill(tc_search_index *index, const float *queries, size_t nq, float radius, const uint64_t *offsets,
                                      size_t total, uint32_t *idx, float *dist);
void tc_search_index_destroy(tc_search_index *index);

/* ---- voxel_grid_filter ----
 * voxel_grid_filter(&PointCloud<Point3f>, voxel_size) -> Result<PointCloud<Point3f>>
 * (threecrate-algorithms/src/filtering.rs:38-133): one f32 centroid per occupied voxel, keys
 * floor((p - bbox_min) / voxel_size), f64 sums in input order.  The reference's output order is
 * unspecified (HashMap); here voxels come out sorted by (kx, ky, kz).  out: capacity n x 3.
 * n == 0 -> TC_OK; voxel_size <= 0 -> TC_INVALID_DATA; 2^21 or more voxels along one
 * axis of the bbox -> TC_UNSUPPORTED. */
tc_status tc_voxel_grid_filter(tc_context *ctx, const float *xyz, size_t n, float voxel_size,
                               float *out_xyz, size_t *n_out);
tc_status tc_voxel_grid_filter_device(tc_context *ctx, const float *d_xyz, size_t n, float voxel_size,
                                      float *d_out_xyz, size_t *n_out);

/* ---- LiDAR frame streaming ----
 * A bounded queue of host frames in front of the per-frame pipeline
 *   voxel_grid_filter -> estimate_normals(previous frame) -> icp_point_to_plane(current -> previous),
 * modelled on RealtimePipeline (threecrate-algorithms/src/streaming.rs:540-646; BackpressureConfig
 * .max_queue_depth): send() blocks the producer while max_queue_depth frames are waiting, try_send()
 * drops the frame instead (metrics.items_dropped), finish() closes the input, drains the queue, joins
 * the worker and returns one result per consecutive frame pair (init = identity) plus the metrics.
 * The host->device copy of frame i+1 overlaps the kernels of frame i (second HIP stream).  Frames are
 * n x 3 floats (stride_floats = 3) or KITTI records x, y, z, intensity (stride_floats = 4); the
 * caller's buffer is free when send returns.  The context must not be used by other calls while the
 * stream exists.  voxel_size <= 0: no filter.  max_correspondence_distance < 0: None. */
typedef struct tc_frame_stream tc_frame_stream;
typedef struct tc_frame_stream_config {
    size_t max_points;                  /* capacity of one frame */
    size_t max_queue_depth;
    float  voxel_size;
    size_t k_neighbors;
    size_t max_iterations;
    float  max_correspondence_distance;
    float  convergence_threshold;
} tc_frame_stream_config;
typedef struct tc_frame_result {
    float    transformation[7];         /* current frame -> previous frame */
    float    mse;
    uint64_t iterations;
    int32_t  converged;
    int32_t  status;                    /* tc_status of this frame's pipeline */
    uint64_t n_points_in, n_points;     /* before / after the voxel filter */
} tc_frame_result;
typedef struct tc_frame_stream_metrics {    /* RealtimeMetrics, streaming.rs */
    uint64_t items_queued, items_processed, items_dropped, max_depth_seen;
} tc_frame_stream_metrics;
tc_status tc_frame_stream_create(tc_context *ctx, const tc_frame_stream_config *config, tc_frame_stream **out);
tc_status tc_frame_stream_send(tc_frame_stream *s, const float *frame, size_t n, size_t stride_floats);
tc_status tc_frame_stream_try_send(tc_frame_stream *s, const float *frame, size_t n, size_t stride_floats, int *accepted);
tc_status tc_frame_stream_finish(tc_frame_stream *s, tc_frame_result *results, size_t capacity, size_t *n_results,
                                 tc_frame_stream_metrics *metrics);
void      tc_frame_stream_destroy(tc_frame_stream *s);
/* VelodyneKittiBinReader::read (threecrate-io/src/lidar.rs:310-343): 16-byte little-endian records
 * x, y, z, intensity -> n x 3 floats.  out_xyz == NULL: only *n_points is set (size query).  A file
 * size that is not a multiple of 16, or a capacity below the point count: TC_INVALID_DATA. */
tc_status tc_read_kitti_bin(const char *path, float *out_xyz, size_t capacity_points, size_t *n_points);

/* ---- profiling ---- */
/* on: 0 = off, 1 = hipEvents around every kernel, 2 = only around every 37th launch of the dominant
   kernel (icp_correspond_reduce): ~1 % overhead (an event is a ~6 us bubble on the stream), used inside bench.py's timed region,
   3 = no events; the ICP main pass counts its searches instead (tc_debug_counter, TC_COUNTER_ICP_*) */
void   tc_profile_enable(tc_context *ctx, int on);
void   tc_profile_reset(tc_context *ctx);
size_t tc_profile_read(tc_context *ctx, tc_kernel_stat *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* THREECRATE_HIP_H */
