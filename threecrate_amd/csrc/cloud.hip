// cloud.hip -- tc_cloud: a device-resident cloud that OWNS its points, its spatial index and (optionally) its normals
// ("device-resident variants (tc_cloud_upload -> handle) so frames can stay on the GPU").
//
// The reference rebuilds its kd-tree inside every call (estimate_normals: normals.rs:272; icp*: registration.rs:281 / :536),
// so a frame that gets normals and then serves as the target of the next registration is indexed twice.  A handle is indexed
// ONCE: estimate_normals leaves the cell-sorted records and the cell-sorted normals in the handle -- exactly the layout the ICP
// kernels read -- and every later registration against it starts from there (no second build, no normals gather, and the
// inscribed-ball bounds of the first registration are kept too).  Results are bit-identical to the handle-free entry points on
// the same grid; the grid's cell edge differs (one edge serves both uses), which changes speed, never answers.
#include "tc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

struct tc_cloud {
    tc_context *ctx = nullptr;
    size_t n = 0;
    tc::DevBuf xyz;                 // n x 3 f32, owned
    tc::DeviceIndex ix;
    bool indexed = false;
    float factor = 0.0f;            // cell-edge factor the index was built with
    bool has_normals = false;       // ix.normals holds the cell-sorted normals of the current index
    tc::DevBuf normals6;            // n x 6 NormalPoint3f in input order (kept when normals were estimated here or set by the caller)
    bool has_normals6 = false;
};

namespace {

using namespace tc;


// One cell edge for both uses when their wishes are close (k = 16 normals want 1.01 x the point spacing, the 1-NN search of ICP
// 1.13): the index is built with the ICP edge then -- measured on the 1 M-point benchmark cloud the normals lose less on the
// larger edge than fifty ICP iterations lose on the smaller one.  Otherwise the first use decides and a later use with a very
// different wish rebuilds (a k = 64 normals grid is a poor 1-NN grid).
float shared_factor(float want) {
    const float icp = icp_cell_factor();
    return (want > 0.8f * icp && want < 1.25f * icp) ? icp : want;
}

// normals6[orig(p)] = {position, cell-sorted normal of p}: the input-order copy of normals that so far only exist in the
// order of the index about to be replaced
__global__ void __launch_bounds__(256) cloud_unsort_normals_kernel(const float4 *__restrict__ pts, const float4 *__restrict__ sorted_nrm,
                                                                  const float *__restrict__ xyz, uint32_t n, float *__restrict__ out6) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t orig = __float_as_uint(pts[p].w);
    const float4 nr = sorted_nrm[p];
    float *o = out6 + 6 * (size_t)orig;
    o[0] = xyz[3 * (size_t)orig]; o[1] = xyz[3 * (size_t)orig + 1]; o[2] = xyz[3 * (size_t)orig + 2];
    o[3] = nr.x; o[4] = nr.y; o[5] = nr.z;
}

tc_status ensure_index(tc_cloud *c, float want_factor, float target_ppo, float min_h) {
    tc_context *ctx = c->ctx;
    const float f = shared_factor(want_factor);
    if (c->indexed && c->factor > 0.8f * f && c->factor < 1.25f * f && min_h <= c->ix.geom.h) return TC_OK;
    if (c->has_normals && !c->has_normals6 && c->indexed && c->n) {
        // The handle's normals live only in the order of the index that is about to be rebuilt (estimated without an output
        // array: tc_cloud_estimate_normals(c, cfg, NULL), the frame stream): keep them in input order first, or the next
        // registration against this handle would gather from an array that was never written.
        if (tc_status s = ensure(ctx, c->normals6, c->n * 6 * sizeof(float))) return s;
        hipLaunchKernelGGL(cloud_unsort_normals_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float4 *)c->ix.pts.p, (const float4 *)c->ix.normals.p, (const float *)c->xyz.p, (uint32_t)c->n,
                           (float *)c->normals6.p);
        TC_HIP_TRY(ctx, hipGetLastError());
        c->has_normals6 = true;
    }
    c->has_normals = false;
    if (tc_status s = build_index(ctx, c->ix, (const float *)c->xyz.p, c->n, f, nullptr, nullptr, nullptr, min_h, target_ppo)) return s;
    c->indexed = true;
    c->factor = f;
    return TC_OK;
}

tc_status cloud_create(tc_context *ctx, const float *p, size_t n, bool from_host, tc_cloud **out) {
    if (!ctx || !out) return TC_INVALID_DATA;
    *out = nullptr;
    if (n >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    tc_cloud *c = new tc_cloud();
    c->ctx = ctx;
    c->n = n;
    if (n) {
        tc_status s = ensure(ctx, c->xyz, n * 3 * sizeof(float));
        if (s == TC_OK && hipMemcpyAsync(c->xyz.p, p, n * 3 * sizeof(float), from_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                                         ctx->stream) != hipSuccess) s = fail(ctx, TC_GPU, "cloud upload failed");
        if (s == TC_OK && from_host && hipStreamSynchronize(ctx->stream) != hipSuccess) s = fail(ctx, TC_GPU, "cloud upload failed");   // the caller's buffer is free on return
        if (s != TC_OK) { recycle(ctx, c->xyz); delete c; return s; }
    }
    *out = c;
    return TC_OK;
}

// sorted float4 normals from an n x stride array in input order
tc_status adopt_normals(tc_cloud *c, const float *d_normals, size_t stride) {
    if (tc_status s = ensure_index(c, icp_cell_factor(), 2.5f, 0.0f)) return s;
    if (tc_status s = gather_normals(c->ctx, c->ix, d_normals, stride)) return s;
    c->has_normals = true;
    return TC_OK;
}

tc_status prepare_target(tc_cloud *tgt, bool p2plane);

tc_status cloud_icp(tc_cloud *src, tc_cloud *tgt, bool p2plane, const float init[7], size_t max_iters, float max_dist, float conv_thr,
                    tc_icp_result *res) {
    if (!src || !tgt || !res || !init) return TC_INVALID_DATA;
    tc_context *ctx = tgt->ctx;
    if (src->ctx != ctx) return fail(ctx, TC_INVALID_DATA, "source and target handles belong to different contexts");
    // validation in the reference's order (registration.rs:266-276 / :517-531)
    if (src->n == 0 || tgt->n == 0) return fail(ctx, TC_INVALID_DATA, "Source or target point cloud is empty");
    if (p2plane && !tgt->has_normals && !tgt->has_normals6)
        return fail(ctx, TC_INVALID_DATA, "target_normals length must equal the number of target points (the target handle has no normals: "
                                          "tc_cloud_estimate_normals or tc_cloud_set_normals_device first)");
    if (max_iters == 0) return fail(ctx, TC_INVALID_DATA, "Max iterations must be positive");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = prepare_target(tgt, p2plane)) return s;
    // (a source handle that has been indexed -- for its own normals -- is walked in that order: no second sort of the source)
    return icp_run(ctx, p2plane, (const float *)src->xyz.p, src->n, (const float *)tgt->xyz.p, tgt->n, nullptr, 0, init, max_iters, max_dist,
                   conv_thr, res, true, 0, &tgt->ix, (src->indexed && src != tgt) ? &src->ix : nullptr);
}

// the target side of a registration against a handle: index (built once), cell-sorted normals
tc_status prepare_target(tc_cloud *tgt, bool p2plane) {
    tc_context *ctx = tgt->ctx;
    if (tc_status s = ensure_index(tgt, icp_cell_factor(), 2.5f, 0.0f)) return s;
    if (p2plane && !tgt->has_normals) {          // the index was rebuilt since the normals were made: re-sort them
        if (!tgt->has_normals6) return fail(ctx, TC_INVALID_DATA, "the target handle has no normals in input order to re-sort after its index was rebuilt");
        if (tc_status s = gather_normals(ctx, tgt->ix, (const float *)tgt->normals6.p + 3, 6)) return s;
        tgt->has_normals = true;
    }
    return TC_OK;
}

}  // namespace

extern "C" {

// One registration over the ranks of a communicator against a TARGET HANDLE (every rank holds the same cloud in its own handle:
// index, normals and inscribed-ball bounds are built once per handle, not once per call -- a map that many scans are registered
// against).  The source is a plain device buffer, sharded as in tc_sharded_icp_point_to_plane_device.
tc_status tc_cloud_sharded_icp(tc_comm *comm, int shard_mode, int point_to_plane, const float *d_source, size_t n_source, tc_cloud *target,
                               const float init[7], size_t max_iters, float max_dist, float conv_thr, tc_icp_result *result) try {
    if (!comm || !target || !result || !init) return TC_INVALID_DATA;
    tc_context *ctx = target->ctx;
    if (comm->ctx != ctx) return fail(ctx, TC_INVALID_DATA, "the communicator belongs to another context");
    if (shard_mode != TC_SHARD_SPATIAL && shard_mode != TC_SHARD_LOCAL && shard_mode != TC_SHARD_INDEX) return fail(ctx, TC_INVALID_DATA, "unknown shard mode");
    const bool may_be_empty = shard_mode == TC_SHARD_LOCAL && comm->nranks > 1;
    if ((n_source == 0 && !may_be_empty) || target->n == 0) return fail(ctx, TC_INVALID_DATA, "Source or target point cloud is empty");
    if (point_to_plane && !target->has_normals && !target->has_normals6)
        return fail(ctx, TC_INVALID_DATA, "target_normals length must equal the number of target points (the target handle has no normals)");
    if (max_iters == 0) return fail(ctx, TC_INVALID_DATA, "Max iterations must be positive");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = prepare_target(target, point_to_plane != 0)) return s;
    return icp_run_sharded(ctx, comm, shard_mode, point_to_plane != 0, d_source, n_source, (const float *)target->xyz.p, target->n, nullptr, 0, init,
                           max_iters, max_dist, conv_thr, result, &target->ix);
} TC_CATCH_STATUS((comm ? comm->ctx : nullptr))

tc_status tc_cloud_upload(tc_context *ctx, const float *xyz, size_t n, tc_cloud **out) try { return cloud_create(ctx, xyz, n, true, out); } TC_CATCH_STATUS(ctx)
tc_status tc_cloud_upload_device(tc_context *ctx, const float *d_xyz, size_t n, tc_cloud **out) try { return cloud_create(ctx, d_xyz, n, false, out); } TC_CATCH_STATUS(ctx)
size_t tc_cloud_size(const tc_cloud *c) { return c ? c->n : 0; }
const float *tc_cloud_points_device(const tc_cloud *c) { return c ? (const float *)c->xyz.p : nullptr; }
// The input-order N x 6 copy is kept by the host-output variant of tc_cloud_estimate_normals; after the device-output variant,
// a NULL output or tc_cloud_set_normals_device the handle holds the cell-sorted normals only and the copy is made here on demand.
const float *tc_cloud_normals_device(const tc_cloud *cc) try {
    tc_cloud *c = const_cast<tc_cloud *>(cc);
    if (!c || c->n == 0) return nullptr;
    if (c->has_normals6) return (const float *)c->normals6.p;
    if (!c->has_normals || !c->indexed) return nullptr;
    // (From here on the call is NOT read-only despite the const handle: it allocates the input-order copy, launches a kernel on the
    // context's stream and waits for it -- not thread-safe against other calls on the same context.  NULL then means "failed", and
    // the context's last error says why; "the handle has no normals" returns NULL above without touching the message.)
    tc_context *ctx = c->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) { (void)fail(ctx, TC_GPU, "tc_cloud_normals_device: hipSetDevice failed"); return nullptr; }
    if (ensure(ctx, c->normals6, c->n * 6 * sizeof(float)) != TC_OK) return nullptr;          // (ensure has set the message)
    hipLaunchKernelGGL(cloud_unsort_normals_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float4 *)c->ix.pts.p, (const float4 *)c->ix.normals.p, (const float *)c->xyz.p, (uint32_t)c->n,
                       (float *)c->normals6.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)fail(ctx, TC_GPU, std::string("tc_cloud_normals_device: ") + hipGetErrorString(e)); return nullptr; }
    c->has_normals6 = true;
    return (const float *)c->normals6.p;
} TC_CATCH_VALUE(nullptr)

static tc_status cloud_normals(tc_cloud *c, const tc_normal_config *cfg, float *out, bool out_on_host, bool keep6) {
    if (!c || !cfg) return TC_INVALID_DATA;
    tc_context *ctx = c->ctx;
    if (c->n == 0) return TC_OK;                                                              // normals.rs:261-263
    if (cfg->k_neighbors < 3) return fail(ctx, TC_INVALID_DATA, "k_neighbors must be at least 3");   // :265-269
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float min_h = cfg->has_radius ? cfg->radius * 0.5005f : 0.0f;
    // the handle's previous normals are superseded from here on: should the index be rebuilt for this k, ensure_index must not
    // spend an unsort kernel and an n x 24 B block on recovering them
    c->has_normals = false;
    c->has_normals6 = false;
    if (tc_status s = ensure_index(c, normals_cell_factor(cfg->k_neighbors, c->n >= kAdaptMinPoints), normals_target_ppo(cfg->k_neighbors), min_h)) return s;
    if (tc_status s = ensure(ctx, c->ix.normals, c->n * sizeof(float4))) return s;
    // the N x 6 records in input order (24-byte scattered stores) are only produced when somebody wants them
    // (a caller's DEVICE array is written directly: no copy of the 24-byte records through the handle's own array -- should the
    // index ever be rebuilt, ensure_index recovers the input-order normals from the cell-sorted ones)
    const bool direct = out != nullptr && !out_on_host && !keep6;
    const bool want6 = (out != nullptr || keep6) && !direct;
    float *d_out6 = direct ? out : nullptr;
    if (want6) {
        if (tc_status s = ensure(ctx, c->normals6, c->n * 6 * sizeof(float))) return s;
        d_out6 = (float *)c->normals6.p;
    }
    c->has_normals = false;
    c->has_normals6 = false;
    // (+ the inscribed-ball bounds of the cloud as an ICP target: they fall out of the same k-NN lists)
    if (tc_status s = normals_on_index(ctx, c->ix, false, 0.0f, (const float *)c->xyz.p, c->n, cfg, d_out6, 0, (size_t)-1, false,
                                       (float4 *)c->ix.normals.p, true)) return s;
    c->has_normals = true;
    c->has_normals6 = want6;
    if (out && !direct) TC_HIP_TRY(ctx, hipMemcpyAsync(out, d_out6, c->n * 6 * sizeof(float), out_on_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
}

tc_status tc_cloud_estimate_normals(tc_cloud *c, const tc_normal_config *cfg, float *out) try { return cloud_normals(c, cfg, out, true, false); } TC_CATCH_STATUS(c ? c->ctx : nullptr)
tc_status tc_cloud_estimate_normals_device(tc_cloud *c, const tc_normal_config *cfg, float *d_out) try { return cloud_normals(c, cfg, d_out, false, false); } TC_CATCH_STATUS(c ? c->ctx : nullptr)

tc_status tc_cloud_set_normals_device(tc_cloud *c, const float *d_normals, size_t n_normals, size_t stride) try {
    if (!c) return TC_INVALID_DATA;
    tc_context *ctx = c->ctx;
    if (n_normals != c->n) return fail(ctx, TC_INVALID_DATA, "target_normals length must equal the number of target points");   // registration.rs:522-526
    if (stride < 3) return fail(ctx, TC_INVALID_DATA, "normal_stride must be >= 3");
    if (c->n == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the handle's previous normals (estimated or set) are superseded: were has_normals left set, an index rebuild inside
    // adopt_normals would "recover" the OLD normals into the input-order copy and mark them valid
    c->has_normals = false;
    c->has_normals6 = false;
    if (tc_status s = adopt_normals(c, d_normals, stride)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the caller's array is free on return
    return TC_OK;
} TC_CATCH_STATUS((c ? c->ctx : nullptr))

tc_status tc_cloud_icp_point_to_plane(tc_cloud *source, tc_cloud *target, const float init[7], size_t max_iters, float max_dist,
                                      float conv_thr, tc_icp_result *result) try {
    return cloud_icp(source, target, true, init, max_iters, max_dist, conv_thr, result);
} TC_CATCH_STATUS((source ? source->ctx : nullptr))

tc_status tc_cloud_icp_detailed(tc_cloud *source, tc_cloud *target, const float init[7], size_t max_iters, float max_dist,
                                float conv_thr, tc_icp_result *result) try {
    return cloud_icp(source, target, false, init, max_iters, max_dist, conv_thr, result);
} TC_CATCH_STATUS((source ? source->ctx : nullptr))

void tc_cloud_destroy(tc_cloud *c) try {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    // the blocks go back to the context's pool: the next frame's handle takes them without a hipMalloc
    tc::recycle(c->ctx, c->xyz); tc::recycle(c->ctx, c->normals6);
    tc::recycle_index(c->ctx, c->ix);
    delete c;
} TC_CATCH_VOID

}  // extern "C"
