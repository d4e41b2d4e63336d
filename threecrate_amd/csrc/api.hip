// api.hip -- C ABI entry points of libthreecrate_hip: context, validation in the reference's
// order and precedence, host<->device staging, error mapping.  No CPU fallback exists: every
// compute entry point needs a HIP device and returns TC_GPU otherwise.
#include "tc_internal.h"
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

namespace tc {

// TC_DEBUG in the SHIPPED library: only the bits that print or count (8 the counting instantiation of the main pass, 64 refine
// statistics, 256 the grid decisions, 1024 block stamps) -- none of them changes a result or the road to it.  The bits that do
// (1 no warm start, 4 no inscribed-ball test, 16 no sums, 32 transform frozen, 512 no edge adaptation, 2048 exact box only, 8192 shells
// only: timing experiments and A/B switches) exist only in the development build, `make dev` ->
// variants/libthreecrate_hip_dev.so (this file compiled with -DTC_DEV, every other object shared with the shipped library): an
// inherited environment variable cannot turn the product into a wrong-answer build (tests/test_debug_bits.py).
[[maybe_unused]] constexpr int kDebugPrintOnlyBits = 8 | 64 | 256 | 1024;
int debug_flags() {
    static const int flags = [] {
        const char *e = getenv("TC_DEBUG");
        const int f = e ? atoi(e) : 0;
#ifdef TC_DEV
        return f;
#else
        return f & kDebugPrintOnlyBits;
#endif
    }();
    return flags;
}

tc_status fail(tc_context *ctx, tc_status st, const std::string &msg) {
    fault_point("fail");
    if (ctx) ctx->last_error = msg;
    return st;
}

// The handlers of the C ABI's function-try-blocks (TC_CATCH_STATUS): must not throw themselves.  The context's message buffer is
// reserved at creation (256 bytes), so that storing a short message allocates nothing; if even that fails the old message stays.
tc_status fail_nothrow(tc_context *ctx, tc_status st, const char *msg) noexcept {
    if (ctx) {
        try { ctx->last_error.assign(msg ? msg : "", std::min<size_t>(msg ? std::strlen(msg) : 0, 240)); } catch (...) { }
    }
    return st;
}

// Fault injection for the tests of those handlers (tests/test_abi_exceptions.py): TC_FAULT=<site>[,<site>...] makes the named sites
// throw std::bad_alloc -- "fail" (every error return that builds a message), "context" (tc_context_create), "batch_thread" (the
// second worker of tc_batch_icp fails to start), "stream_worker" (the frame streamer's thread body), "kitti" (tc_read_kitti_bin).
// Read per call: a test sets it after the library has been loaded.  DEVELOPMENT BUILD ONLY (-DTC_DEV, `make dev`): in the shipped
// library this is an empty function -- an inherited TC_FAULT cannot turn error returns into throws, and no worker thread calls
// getenv next to a host that may be changing its environment.
void fault_point(const char *site) {
#ifdef TC_DEV
    const char *e = getenv("TC_FAULT");
    if (!e || !*e) return;
    const size_t n = std::strlen(site);
    for (const char *p = e; *p;) {
        const char *q = std::strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : std::strlen(p);
        if (len == n && std::strncmp(p, site, n) == 0) throw std::bad_alloc();
        p = q ? q + 1 : p + len;
    }
#else
    (void)site;            // the shipped library never reads TC_FAULT (no getenv on worker threads, no injected throw)
#endif
}

// The pool parks what destroyed handles give back so that a handle per frame costs no hipMalloc.  Its cap follows the blocks it has
// seen -- a few handles' worth (a handle holds about a dozen buffers of at most `pool_largest` bytes) -- instead of a flat 32 GB:
// memory parked here is invisible to every other allocator of the process (torch's caching allocator, RCCL, a second context),
// which would run out of memory next to it.  tc_context_trim releases it on demand.
constexpr size_t kPoolMaxBytes = (size_t)32 << 30, kPoolMinBytes = (size_t)256 << 20, kPoolMaxBlocks = 96;

void recycle(tc_context *ctx, DevBuf &b) {
    if (!b.p) return;
    ctx->pool_largest = std::max(ctx->pool_largest, b.cap);
    const size_t cap_bytes = std::min(kPoolMaxBytes, std::max(kPoolMinBytes, 48 * ctx->pool_largest));
    if (ctx->pool.size() >= kPoolMaxBlocks || ctx->pool_bytes + b.cap > cap_bytes) {
        (void)hipFree(b.p);
    } else {
        ctx->pool.push_back(b);
        ctx->pool_bytes += b.cap;
    }
    b.p = nullptr; b.cap = 0;
}

tc_status ensure(tc_context *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return TC_OK;
    if (b.p) {
        // buffers may still be referenced by work in flight on the stream
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, TC_GPU, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
        recycle(ctx, b);
    }
    // best fit from the pool: the smallest block that holds the request without wasting more than the request itself
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i) {
        const size_t cap = ctx->pool[i].cap;
        if (cap >= bytes && cap <= 2 * bytes + 4096 && (best < 0 || cap < ctx->pool[best].cap)) best = i;
    }
    if (best >= 0) {
        b = ctx->pool[best];
        ctx->pool_bytes -= b.cap;
        ctx->pool.erase(ctx->pool.begin() + best);
        return TC_OK;
    }
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess && !ctx->pool.empty()) {          // out of memory with blocks parked in the pool: release them, try again
        (void)hipGetLastError();
        for (auto &pb : ctx->pool) (void)hipFree(pb.p);
        ctx->pool.clear(); ctx->pool_bytes = 0;
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, TC_GPU, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.cap = want;
    return TC_OK;
}

ProfScope::ProfScope(tc_context *c, const char *name, bool dominant) : ctx(c) {
    if (ctx->profiling != 1 && ctx->profiling != 2) return;          // (3 = search statistics: no events)
    // (mode 2: every 37th launch of the dominant kernel (17th until round 6: the bubbles were 36 us of a 2.8 ms step) -- a stride coprime to the usual 50 iterations per call, so that the sampled
    // iteration indices walk through all of 0 .. 49 over the calls and the cold first pass is sampled as often as any other.  An event on the stream is a ~5.7 us bubble on either side of the kernel --
    // every 4th launch, as until round 4, was 2.9 us per ICP iteration = 5 % of the timed region, not the 1 % once estimated.)
    if (ctx->profiling == 2 && (!dominant || (ctx->prof_tick++ % 37u) != 0)) return;
    for (size_t i = 0; i < ctx->timers.size(); ++i)
        if (ctx->timers[i].name == name) { idx = (int)i; break; }
    if (idx < 0) { ctx->timers.push_back(KernelTimer{name, {}, 0, 0.0, 1e300, 0.0}); idx = (int)ctx->timers.size() - 1; }
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    e0 = get(); e1 = get();
    ext = dominant;                                   // (the launch records both events itself: see the struct)
    if (!ext) (void)hipEventRecord(e0, ctx->stream);
}
ProfScope::~ProfScope() {
    if (idx < 0) return;
    if (!ext) (void)hipEventRecord(e1, ctx->stream);
    ctx->timers[idx].pending.emplace_back(e0, e1);
}

bool pinned_poll_enabled() {
    // (read per call like the other switches: a test or tools/dev/paths_stress.py sets it after the library has already run)
    const char *e = getenv("TC_NO_PINNED_POLL");
    return !(e && atoi(e) != 0);
}

void *pinned_dev_ptr(tc_context *ctx, const void *host_addr) {
    if (!ctx->pinned_dev) {
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, ctx->pinned, 0) != hipSuccess) return nullptr;
        ctx->pinned_dev = d;
    }
    return (char *)ctx->pinned_dev + ((const char *)host_addr - (const char *)ctx->pinned);
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    sched_yield();
#endif
}

// Three regimes: the words that arrive within a kernel or two (bounding box, occupancy: ~10-100 us) are caught spinning; a wait
// that outlasts ~4 k spins yields the core between looks (the ranks of one host share its cores with each other and with
// RCCL's proxy threads); a wait of more than ~2 ms -- the tail of a long registration -- sleeps 50 us between looks instead of
// keeping a core busy for tens of milliseconds (the overshoot is noise at that length).
tc_status wait_pinned_word(tc_context *ctx, volatile uint32_t *word, const char *what) {
    struct timespec t0 = {0, 0};
    bool sleeping = false;
    for (unsigned spins = 0; *word == 0u; ++spins) {
        // spinning / yielding: one stream query per 1024 looks; sleeping: one per look (a look is 50 us by then, so a faulted stream
        // or one that drained without writing the word is noticed within a look, not after 50 ms -- ADVICE r5)
        if (sleeping || (spins & 1023u) == 1023u) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) { if (*word == 0u) return fail(ctx, TC_GPU, std::string("internal error: ") + what + ": the stream drained without the word being written"); break; }
            if (q != hipErrorNotReady) return fail(ctx, TC_GPU, std::string(what) + ": " + hipGetErrorString(q));
            if (!sleeping && spins >= 4096u) {
                struct timespec t;
                clock_gettime(CLOCK_MONOTONIC, &t);
                if (t0.tv_sec == 0 && t0.tv_nsec == 0) t0 = t;
                sleeping = (t.tv_sec - t0.tv_sec) * 1000000000ll + (t.tv_nsec - t0.tv_nsec) > 2000000ll;
            }
        }
        if (sleeping) { const struct timespec d = {0, 50000}; nanosleep(&d, nullptr); }
        else if (spins > 4096u) sched_yield();
        else cpu_relax();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return TC_OK;
}

static void free_buf(DevBuf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

// NormalEstimationConfig -> cell edge factor: ring R0 = 2 must cover the (k+1)-NN sphere for all
// but ~1e-3 of the queries of a locally uniform cloud (Poisson tail), the rest take the overflow pass.
tc_status upload_async(tc_context *ctx, void *d_dst, const void *h_src, size_t bytes) {
    // small uploads stay on the context's stream: a few hundred microseconds of copy have nothing to hide under (measured no
    // difference either way on a 230 k-point pair), and one stream less is one thing less to go wrong
    if (bytes < ((size_t)8 << 20)) {
        TC_HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return TC_OK;
    }
    ctx->upload_used_copy_stream = true;
    if (!ctx->copy_stream) TC_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->upload_event) TC_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->upload_event, hipEventDisableTiming));
    // the destination may still be read by work the context's stream holds from an earlier call: entry points return
    // synchronised, so nothing is in flight here
    TC_HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    return TC_OK;
}
tc_status uploads_issued(tc_context *ctx) {
    if (!ctx->upload_used_copy_stream) return TC_OK;
    ctx->upload_used_copy_stream = false;
    TC_HIP_TRY(ctx, hipEventRecord(ctx->upload_event, ctx->copy_stream));
    ctx->upload_pending = true;
    return TC_OK;
}
tc_status wait_uploads(tc_context *ctx) {
    if (!ctx->upload_pending) return TC_OK;
    ctx->upload_pending = false;
    TC_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->upload_event, 0));
    return TC_OK;
}

float normals_cell_factor(size_t k, bool large) {
    const double K1 = (double)k + 1.0;
    const double lam = K1 + 3.1 * std::sqrt(K1) + 2.0;
    const double c = std::cbrt(lam / 4.18879);
    // multiplier scanned on the 1 M-point uniform cloud (k = 16, whole call): 0.8 -> 0.95 ms, 0.9 -> 0.83, 0.95 -> 0.79,
    // 1.0 -> 0.75, 1.1 -> 0.75, 1.2 -> 0.79, 1.3 -> 0.84 (the in-place ring continuation made the overflow tail cheap);
    // again at 7 waves per SIMD: 0.9 -> 0.70, 0.95 -> 0.67, 1.0 -> 0.655, 1.03 -> 0.65, 1.1 -> 0.71
    // after the centre-out pruned list pass (round 2; kernel only, TC_NORMALS_CELL_MULT): 0.8 -> 537 us, 0.9 -> 479, 1.0 -> 445,
    // 1.03 -> 444, 1.1 -> 440, 1.2 -> 442, 1.3 -> 465: flat between 1.0 and 1.2
    // Applied to clouds of >= 2^18 points (those whose edge is also adapted to the measured occupancy); smaller clouds keep
    // 0.95: they are not adapted, and on surface-like frames (24 k-point voxel-filtered LiDAR) the larger edge costs 5 %.
    static const double mult = [] { const char *e = getenv("TC_NORMALS_CELL_MULT"); return e ? atof(e) : 0.0; }();   // (tuning experiments)
    return (float)((mult > 0.0 ? mult : (large ? 1.03 : 0.95)) * c / 2.0);
}

// points per occupied cell wanted on a SURFACE: the disc of radius ~1.9 h (ring 2) must hold the same
// lambda(k) points the volumetric factor above puts into the ring-2 sphere: sigma h^2 = lambda / (pi 1.9^2)
float normals_target_ppo(size_t k) {
    const double K1 = (double)k + 1.0;
    return (float)((K1 + 3.1 * std::sqrt(K1) + 2.0) / 11.3);
}

void recycle_index(tc_context *ctx, DeviceIndex &ix) {
    for (DevBuf *b : {&ix.pts, &ix.cell_start, &ix.normals, &ix.vor, &ix.pts12, &ix.cell_of, &ix.slot, &ix.arrival, &ix.fill, &ix.blocksum}) recycle(ctx, *b);
    ix.pts12_valid = false; ix.vor_valid = false;
}

void free_index(DeviceIndex &ix) {
    free_buf(ix.pts); free_buf(ix.cell_start); free_buf(ix.normals); free_buf(ix.vor); free_buf(ix.pts12); free_buf(ix.cell_of);
    free_buf(ix.slot); free_buf(ix.arrival); free_buf(ix.fill); free_buf(ix.blocksum);
}

// build == false: `ix` already indexes this cloud (a cloud handle)
tc_status normals_on_index(tc_context *ctx, DeviceIndex &ix, bool build, float cell_factor_override, const float *d_xyz, size_t n,
                           const tc_normal_config *cfg, float *d_out, size_t p_begin, size_t p_end, bool slice_out, float4 *d_sorted_nrm,
                           bool with_bounds) {
    if (build) {
        // radius mode: ring 2 must cover the radius ball, so the cell edge is at least radius / 2
        const float min_h = cfg->has_radius ? cfg->radius * 0.5005f : 0.0f;
        const float f = cell_factor_override > 0.0f ? cell_factor_override : normals_cell_factor(cfg->k_neighbors, n >= kAdaptMinPoints);
        // a SLICE of the cell-sorted order is some rank's share of it: every rank has to build the same order (strict_order)
        if (tc_status s = build_index(ctx, ix, d_xyz, n, f, nullptr, nullptr, nullptr, min_h, normals_target_ppo(cfg->k_neighbors), slice_out)) return s;
    }
    float vp[3];
    if (cfg->has_viewpoint) {
        vp[0] = cfg->viewpoint[0]; vp[1] = cfg->viewpoint[1]; vp[2] = cfg->viewpoint[2];
    } else {   // normals.rs:275-303 (f32, same operation order; min/max are order independent)
        const float *bmn = ix.exact_min, *bmx = ix.exact_max;      // the cloud's exact box (the grid's may be clamped)
        const float cx = (bmn[0] + bmx[0]) / 2.0f, cy = (bmn[1] + bmx[1]) / 2.0f, cz = (bmn[2] + bmx[2]) / 2.0f;
        const float ex = bmx[0] - bmn[0], ey = bmx[1] - bmn[1], ez = bmx[2] - bmn[2];
        const float extent = std::sqrt(ex * ex + ey * ey + ez * ez);
        vp[0] = cx + 0.0f; vp[1] = cy + 0.0f; vp[2] = cz + extent;
    }
    // the inscribed-ball bounds of the ICP main pass fall out of the same k-NN lists (k-NN mode, whole cloud, the list must hold
    // the query and at least one other record)
    float4 *d_vor = nullptr;
    const bool knn_mode = !(cfg->has_radius && cfg->radius > 0.0f);
    if (with_bounds && knn_mode && p_begin == 0 && p_end >= n) {
        if (tc_status s = ensure(ctx, ix.vor, n * sizeof(float4))) return s;
        d_vor = (float4 *)ix.vor.p;
    }
    if (tc_status s = launch_normals(ctx, ix, d_xyz, *cfg, vp, d_out, p_begin, p_end, slice_out, d_sorted_nrm, d_vor)) return s;
    ix.vor_valid = d_vor != nullptr;
    return TC_OK;
}

static tc_status normals_device(tc_context *ctx, const float *d_xyz, size_t n, const tc_normal_config *cfg, float *d_out,
                                size_t p_begin = 0, size_t p_end = (size_t)-1, bool slice_out = false) {
    return normals_on_index(ctx, ctx->tgt_index, true, 0.0f, d_xyz, n, cfg, d_out, p_begin, p_end, slice_out, nullptr);
}

}  // namespace tc

using namespace tc;

extern "C" {

int tc_abi_version(void) { return TC_ABI_VERSION; }

int tc_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} TC_CATCH_VALUE(0)

static tc_status context_create(int device, void *stream, bool own, tc_context **out) {
    if (!out) return TC_INVALID_DATA;
    *out = nullptr;
    fault_point("context");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TC_GPU;
    if (hipSetDevice(device) != hipSuccess) return TC_GPU;
    std::unique_ptr<tc_context> holder(new tc_context());          // (an exception below must not leak it)
    tc_context *ctx = holder.get();
    ctx->last_error.reserve(256);          // fail_nothrow stores its messages without allocating
    ctx->device = device;
    ctx->own_stream = own;
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return TC_GPU;
    } else {
        ctx->stream = (hipStream_t)stream;
    }
    ctx->pinned_cap = 1 << 16;
    if (hipHostMalloc(&ctx->pinned, ctx->pinned_cap, hipHostMallocDefault) != hipSuccess) {
        if (own) (void)hipStreamDestroy(ctx->stream);
        return TC_GPU;
    }
    *out = holder.release();
    return TC_OK;
}

tc_status tc_context_create(int device, tc_context **out) try { return context_create(device, nullptr, true, out); } TC_CATCH_STATUS(nullptr)
tc_status tc_context_create_on_stream(int device, void *hip_stream, tc_context **out) try {
    return context_create(device, hip_stream, false, out);
} TC_CATCH_STATUS(nullptr)

tc_status tc_context_wait_stream(tc_context *ctx, void *other_hip_stream) try {
    if (!ctx) return TC_INVALID_DATA;
    hipStream_t other = (hipStream_t)other_hip_stream;
    if (other == ctx->stream) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->order_event) TC_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
    TC_HIP_TRY(ctx, hipEventRecord(ctx->order_event, other));
    TC_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->order_event, 0));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

// the other direction: work enqueued on `other` AFTER this call waits for everything the context's stream holds now (a caller that
// is about to overwrite or free a buffer it just handed to tc_cloud_upload_device: no host wait)
tc_status tc_stream_wait_context(tc_context *ctx, void *other_hip_stream) try {
    if (!ctx) return TC_INVALID_DATA;
    hipStream_t other = (hipStream_t)other_hip_stream;
    if (other == ctx->stream) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->release_event) TC_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->release_event, hipEventDisableTiming));
    TC_HIP_TRY(ctx, hipEventRecord(ctx->release_event, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamWaitEvent(other, ctx->release_event, 0));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

void tc_context_destroy(tc_context *ctx) try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->upload_event) (void)hipEventDestroy(ctx->upload_event);
    free_index(ctx->tgt_index); free_index(ctx->src_index); free_index(ctx->vox_index);
    free_buf(ctx->in_a); free_buf(ctx->in_b); free_buf(ctx->in_c); free_buf(ctx->out_a); free_buf(ctx->bbox);
    free_buf(ctx->state); free_buf(ctx->partials); free_buf(ctx->corr); free_buf(ctx->gicp_src_cov); free_buf(ctx->overflow); free_buf(ctx->normals_hard); free_buf(ctx->build_tmp); free_buf(ctx->dbg_times); free_buf(ctx->icp_wsrc);
    for (auto &pb : ctx->pool) (void)hipFree(pb.p);
    for (auto e : ctx->chunk_events) (void)hipEventDestroy(e);
    if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
    if (ctx->release_event) (void)hipEventDestroy(ctx->release_event);
    for (auto &t : ctx->timers) for (auto &p : t.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
} TC_CATCH_VOID

tc_status tc_context_trim(tc_context *ctx) try {
    if (!ctx) return TC_INVALID_DATA;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &pb : ctx->pool) (void)hipFree(pb.p);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    ctx->pool_largest = 0;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

const char *tc_last_error_message(const tc_context *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

tc_status tc_synchronize(tc_context *ctx) try {
    if (!ctx) return TC_INVALID_DATA;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

void tc_normal_config_default(tc_normal_config *cfg) try {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->k_neighbors = 10;            // normals.rs:28-36
    cfg->has_radius = 0;
    cfg->consistent_orientation = 1;
    cfg->has_viewpoint = 0;
} TC_CATCH_VOID

// ---- normals --------------------------------------------------------------------------------
static tc_status normals_validate(tc_context *ctx, size_t n, const tc_normal_config *cfg, bool *empty) {
    *empty = false;
    if (!ctx || !cfg) return TC_INVALID_DATA;
    if (n == 0) { *empty = true; return TC_OK; }                                        // normals.rs:261-263
    if (cfg->k_neighbors < 3) return fail(ctx, TC_INVALID_DATA, "k_neighbors must be at least 3");   // :265-269
    return TC_OK;
}

tc_status tc_estimate_normals_device(tc_context *ctx, const float *d_xyz, size_t n, const tc_normal_config *cfg,
                                     float *d_out) try {
    bool empty;
    if (tc_status s = normals_validate(ctx, n, cfg, &empty)) return s;
    if (empty) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = normals_device(ctx, d_xyz, n, cfg, d_out)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

// ---- sharded normals of one big cloud over several GPUs ----
tc_status tc_estimate_normals_slice_device(tc_context *ctx, const float *d_xyz, size_t n, const tc_normal_config *cfg, size_t begin,
                                           size_t end, float *d_slice_out) try {
    bool empty;
    if (tc_status s = normals_validate(ctx, n, cfg, &empty)) return s;
    if (empty) return TC_OK;
    if (begin > end || end > n) return fail(ctx, TC_INVALID_DATA, "normals slice: need begin <= end <= n");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = normals_device(ctx, d_xyz, n, cfg, d_slice_out, begin, end, true)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_normals_unsort_device(tc_context *ctx, const float *d_sorted_all, size_t n, float *d_out) try {
    if (!ctx) return TC_INVALID_DATA;
    if (n == 0) return TC_OK;
    if (ctx->tgt_index.geom.n != n || !ctx->tgt_index.pts.p)
        return fail(ctx, TC_INVALID_DATA, "normals unsort: call tc_estimate_normals_slice_device on this context with the same cloud first");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = launch_normals_unsort(ctx, ctx->tgt_index, d_sorted_all, d_out)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_estimate_normals(tc_context *ctx, const float *xyz, size_t n, const tc_normal_config *cfg, float *out) try {
    bool empty;
    if (tc_status s = normals_validate(ctx, n, cfg, &empty)) return s;
    if (empty) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = ensure(ctx, ctx->in_a, n * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->out_a, n * 6 * sizeof(float))) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_a.p, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status s = normals_device(ctx, (const float *)ctx->in_a.p, n, cfg, (float *)ctx->out_a.p)) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->out_a.p, n * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

// ---- ICP ------------------------------------------------------------------------------------
static tc_status icp_validate(tc_context *ctx, size_t ns, size_t nt, size_t max_iters, const tc_icp_result *res) {
    if (!ctx || !res) return TC_INVALID_DATA;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "Source or target point cloud is empty");   // registration.rs:266-270
    if (max_iters == 0) return fail(ctx, TC_INVALID_DATA, "Max iterations must be positive");             // :272-276
    if (ns >= 0xFFFFFFF0ull || nt >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    return TC_OK;
}

tc_status tc_icp_detailed_device(tc_context *ctx, const float *d_source, size_t n_source, const float *d_target,
                                 size_t n_target, const float init[7], size_t max_iters, float max_dist, float conv_thr,
                                 tc_icp_result *result) try {
    if (tc_status s = icp_validate(ctx, n_source, n_target, max_iters, result)) return s;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return icp_run(ctx, false, d_source, n_source, d_target, n_target, nullptr, 0, init, max_iters, max_dist, conv_thr, result, true);
} TC_CATCH_STATUS(ctx)

tc_status tc_icp_detailed(tc_context *ctx, const float *source, size_t n_source, const float *target, size_t n_target,
                          const float init[7], size_t max_iters, float max_dist, float conv_thr, tc_icp_result *result) try {
    if (tc_status s = icp_validate(ctx, n_source, n_target, max_iters, result)) return s;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = ensure(ctx, ctx->in_a, n_source * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->in_b, n_target * 3 * sizeof(float))) return s;
    // the target first, on the context's stream: its index build starts as soon as it has landed; the source follows on the copy
    // stream, under the build (icp_setup waits for it before it orders the source)
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_b.p, target, n_target * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status s = upload_async(ctx, ctx->in_a.p, source, n_source * 3 * sizeof(float))) return s;
    if (tc_status s = uploads_issued(ctx)) return s;
    const tc_status rc = icp_run(ctx, false, (const float *)ctx->in_a.p, n_source, (const float *)ctx->in_b.p, n_target, nullptr, 0, init,
                                 max_iters, max_dist, conv_thr, result, false);
    if (ctx->upload_pending) { ctx->upload_pending = false; (void)hipStreamSynchronize(ctx->copy_stream); }     // (an early error return)
    return rc;
} TC_CATCH_STATUS(ctx)

tc_status tc_icp_point_to_point(tc_context *ctx, const float *source, size_t n_source, const float *target, size_t n_target,
                                const float init[7], size_t max_iterations, float conv_thr, float max_dist,
                                tc_icp_result *result) try {
    if (tc_status s = icp_validate(ctx, n_source, n_target, max_iterations, result)) return s;
    if (!(conv_thr > 0.0f)) return fail(ctx, TC_INVALID_DATA, "Convergence threshold must be positive");   // registration.rs:665-669
    return tc_icp_detailed(ctx, source, n_source, target, n_target, init, max_iterations, max_dist, conv_thr, result);
} TC_CATCH_STATUS(ctx)

tc_status tc_icp(tc_context *ctx, const float *source, size_t n_source, const float *target, size_t n_target,
                 const float init[7], size_t max_iters, float out[7]) try {
    if (!ctx || !out || !init) return TC_INVALID_DATA;
    tc_icp_result r;
    std::memset(&r, 0, sizeof(r));
    tc_status s = tc_icp_detailed(ctx, source, n_source, target, n_target, init, max_iters, -1.0f, 1e-6f, &r);   // registration.rs:238
    if (s == TC_OK) std::memcpy(out, r.transformation, 7 * sizeof(float));
    else std::memcpy(out, init, 7 * sizeof(float));                                                             // :240
    return TC_OK;
} TC_CATCH_STATUS(ctx)

static tc_status p2plane_validate(tc_context *ctx, size_t ns, size_t nt, size_t nn, size_t stride, size_t max_iters,
                                  const tc_icp_result *res) {
    if (!ctx || !res) return TC_INVALID_DATA;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "Source or target point cloud is empty");            // registration.rs:517-521
    if (nn != nt) return fail(ctx, TC_INVALID_DATA, "target_normals length must equal the number of target points"); // :522-526
    if (max_iters == 0) return fail(ctx, TC_INVALID_DATA, "Max iterations must be positive");                        // :527-531
    if (stride < 3) return fail(ctx, TC_INVALID_DATA, "normal_stride must be >= 3");
    if (ns >= 0xFFFFFFF0ull || nt >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    return TC_OK;
}

tc_status tc_icp_point_to_plane_detailed_device(tc_context *ctx, const float *d_source, size_t n_source,
                                                const float *d_target, size_t n_target, const float *d_normals,
                                                size_t n_normals, size_t stride, const float init[7], size_t max_iters,
                                                float max_dist, float conv_thr, tc_icp_result *result) try {
    if (tc_status s = p2plane_validate(ctx, n_source, n_target, n_normals, stride, max_iters, result)) return s;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return icp_run(ctx, true, d_source, n_source, d_target, n_target, d_normals, stride, init, max_iters, max_dist, conv_thr,
                   result, true);
} TC_CATCH_STATUS(ctx)

tc_status tc_icp_point_to_plane_detailed(tc_context *ctx, const float *source, size_t n_source, const float *target,
                                         size_t n_target, const float *normals, size_t n_normals, size_t stride,
                                         const float init[7], size_t max_iters, float max_dist, float conv_thr,
                                         tc_icp_result *result) try {
    if (tc_status s = p2plane_validate(ctx, n_source, n_target, n_normals, stride, max_iters, result)) return s;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t nbytes = ((n_normals - 1) * stride + 3) * sizeof(float);
    if (tc_status s = ensure(ctx, ctx->in_a, n_source * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->in_b, n_target * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->in_c, nbytes)) return s;
    // the target first, on the context's stream: its index build starts as soon as it has landed; normals and source follow on
    // the copy stream, under the build (icp_setup waits for them before it gathers the normals)
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_b.p, target, n_target * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status s = upload_async(ctx, ctx->in_c.p, normals, nbytes)) return s;
    if (tc_status s = upload_async(ctx, ctx->in_a.p, source, n_source * 3 * sizeof(float))) return s;
    if (tc_status s = uploads_issued(ctx)) return s;
    const tc_status rc = icp_run(ctx, true, (const float *)ctx->in_a.p, n_source, (const float *)ctx->in_b.p, n_target,
                                 (const float *)ctx->in_c.p, stride, init, max_iters, max_dist, conv_thr, result, false);
    if (ctx->upload_pending) { ctx->upload_pending = false; (void)hipStreamSynchronize(ctx->copy_stream); }     // (an early error return)
    return rc;
} TC_CATCH_STATUS(ctx)

// ---- one registration / one cloud over the ranks of a communicator ----------------------------------
tc_status tc_sharded_icp_point_to_plane_device(tc_context *ctx, tc_comm *comm, int shard_mode, const float *d_source, size_t n_source,
                                               const float *d_target, size_t n_target, const float *d_normals, size_t n_normals,
                                               size_t stride, const float init[7], size_t max_iters, float max_dist, float conv_thr,
                                               tc_icp_result *result) try {
    if (!ctx || !comm || !result) return TC_INVALID_DATA;
    if (comm->ctx != ctx) return fail(ctx, TC_INVALID_DATA, "the communicator belongs to another context");
    if (shard_mode != TC_SHARD_SPATIAL && shard_mode != TC_SHARD_LOCAL && shard_mode != TC_SHARD_INDEX) return fail(ctx, TC_INVALID_DATA, "unknown shard mode");
    // a rank of a TC_SHARD_LOCAL run may own no source points (the other ranks do)
    const size_t ns_check = (shard_mode == TC_SHARD_LOCAL && comm->nranks > 1 && n_source == 0) ? 1 : n_source;
    if (tc_status s = p2plane_validate(ctx, ns_check, n_target, n_normals, stride, max_iters, result)) return s;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return icp_run_sharded(ctx, comm, shard_mode, true, d_source, n_source, d_target, n_target, d_normals, stride, init, max_iters, max_dist,
                           conv_thr, result);
} TC_CATCH_STATUS(ctx)

tc_status tc_sharded_icp_detailed_device(tc_context *ctx, tc_comm *comm, int shard_mode, const float *d_source, size_t n_source,
                                         const float *d_target, size_t n_target, const float init[7], size_t max_iters, float max_dist,
                                         float conv_thr, tc_icp_result *result) try {
    if (!ctx || !comm || !result) return TC_INVALID_DATA;
    if (comm->ctx != ctx) return fail(ctx, TC_INVALID_DATA, "the communicator belongs to another context");
    if (shard_mode != TC_SHARD_SPATIAL && shard_mode != TC_SHARD_LOCAL && shard_mode != TC_SHARD_INDEX) return fail(ctx, TC_INVALID_DATA, "unknown shard mode");
    const size_t ns_check = (shard_mode == TC_SHARD_LOCAL && comm->nranks > 1 && n_source == 0) ? 1 : n_source;
    if (tc_status s = icp_validate(ctx, ns_check, n_target, max_iters, result)) return s;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return icp_run_sharded(ctx, comm, shard_mode, false, d_source, n_source, d_target, n_target, nullptr, 0, init, max_iters, max_dist,
                           conv_thr, result);
} TC_CATCH_STATUS(ctx)

// this rank's slot of ceil(n / W) cell-sorted positions -> slot_out (rows x 6); the fallible part of the sharded normals
static tc_status sharded_normals_slot(tc_context *ctx, tc_comm *comm, const float *d_xyz, size_t n, const tc_normal_config *cfg,
                                      float *slot_out, size_t &lo, size_t &hi) {
    const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank;
    const size_t rows = (n + W - 1) / W;
    lo = std::min(r * rows, n); hi = std::min((r + 1) * rows, n);
    return normals_device(ctx, d_xyz, n, cfg, slot_out, lo, hi, true);
}

tc_status tc_sharded_estimate_normals_device(tc_context *ctx, tc_comm *comm, const float *d_xyz, size_t n, const tc_normal_config *cfg,
                                             float *d_out) try {
    if (!comm) return TC_INVALID_DATA;
    bool empty;
    if (tc_status s = normals_validate(ctx, n, cfg, &empty)) return s;
    if (empty) return TC_OK;
    if (comm->ctx != ctx) return fail(ctx, TC_INVALID_DATA, "the communicator belongs to another context");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t W = (size_t)comm->nranks, r = (size_t)comm->rank;
    // equal slots of `rows` records (the last ranks' ranges may be shorter or empty): the all-gather runs in place
    const size_t rows = (n + W - 1) / W;
    // everything that can fail on ONE rank (allocation, index build, launches) happens before the collective, and the ranks agree
    // on it: a rank returning early would leave its peers waiting in the all-gather for ever (comm_agree)
    tc_status local = ensure(ctx, ctx->out_a, W * rows * 6 * sizeof(float));
    float *sorted_all = (float *)ctx->out_a.p;
    size_t lo = 0, hi = 0;
    if (local == TC_OK) local = sharded_normals_slot(ctx, comm, d_xyz, n, cfg, sorted_all + r * rows * 6, lo, hi);
    if (tc_status s = comm_agree(comm, local)) return s;
    if (tc_status s = comm_allgather(comm, sorted_all, rows * 6 * sizeof(float))) return s;
    // slot q holds the cell-sorted positions [q rows, min((q + 1) rows, n)): the slots are contiguous in position
    if (tc_status s = launch_normals_unsort(ctx, ctx->tgt_index, sorted_all, d_out)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

__global__ void __launch_bounds__(256) slice_orig_index_kernel(const float4 *__restrict__ pts, uint32_t lo, uint32_t count, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = __float_as_uint(pts[lo + i].w);
}

tc_status tc_sharded_estimate_normals_local_device(tc_context *ctx, tc_comm *comm, const float *d_xyz, size_t n, const tc_normal_config *cfg,
                                                   float *d_out_slice, uint32_t *d_orig_index, size_t *first, size_t *count) try {
    if (!comm || !first || !count) return TC_INVALID_DATA;
    *first = 0; *count = 0;
    bool empty;
    if (tc_status s = normals_validate(ctx, n, cfg, &empty)) return s;
    if (empty) return TC_OK;
    if (comm->ctx != ctx) return fail(ctx, TC_INVALID_DATA, "the communicator belongs to another context");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t lo = 0, hi = 0;
    if (tc_status s = sharded_normals_slot(ctx, comm, d_xyz, n, cfg, d_out_slice, lo, hi)) return s;
    if (d_orig_index && hi > lo) {
        hipLaunchKernelGGL(slice_orig_index_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float4 *)ctx->tgt_index.pts.p, (uint32_t)lo, (uint32_t)(hi - lo), d_orig_index);
        TC_HIP_TRY(ctx, hipGetLastError());
    }
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *first = lo; *count = hi - lo;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

unsigned long long tc_debug_counter(const tc_context *ctx, int which) try {
    if (!ctx) return 0;
    if (which >= TC_COUNTER_ICP_ITERATIONS && which <= TC_COUNTER_ICP_STEPS_TAKEN) return ctx->stat_icp[which - TC_COUNTER_ICP_ITERATIONS];
    return which == TC_COUNTER_INDEXED_POINTS ? ctx->stat_indexed_points : which == TC_COUNTER_INDEX_BUILDS ? ctx->stat_index_builds : 0ull;
} TC_CATCH_VALUE(0)

tc_status tc_batch_icp(tc_context *const *ctxs, size_t n_ctx, const tc_batch_icp_job *jobs, size_t n_jobs,
                       tc_batch_icp_result *results) try {
    if (!ctxs || n_ctx == 0 || (!jobs && n_jobs) || (!results && n_jobs)) return TC_INVALID_DATA;
    static const float identity[7] = {0, 0, 0, 1, 0, 0, 0};   // gpu/icp.rs:202: always starts from identity
    auto worker = [&](size_t c) {
        for (size_t j = c; j < n_jobs; j += n_ctx) {
            tc_icp_result r;
            std::memset(&r, 0, sizeof(r));
            const tc_batch_icp_job &jb = jobs[j];
            tc_status s = tc_icp_point_to_point(ctxs[c], jb.source, jb.n_source, jb.target, jb.n_target, identity,
                                                jb.max_iterations, jb.convergence_threshold, jb.max_correspondence_distance, &r);
            std::memcpy(results[j].transformation, s == TC_OK ? r.transformation : identity, 7 * sizeof(float));
            results[j].final_error = r.mse;
            results[j].iterations = r.iterations;
            results[j].status = (int32_t)s;
        }
    };
    if (n_ctx == 1) { worker(0); return TC_OK; }
    // One thread per context.  A thread that cannot be started (std::system_error, std::bad_alloc) must not take the started ones
    // down with it -- destroying a joinable std::thread is std::terminate --: the contexts left without a thread are served on the
    // caller's thread, one after the other, and every started thread is joined.  (worker() itself cannot throw: it calls wrapped
    // entry points and copies plain structs.)
    std::vector<std::thread> th;
    size_t started = 0;
    try {
        th.reserve(n_ctx);
        for (; started < n_ctx; ++started) {
            if (started == 1) fault_point("batch_thread");
            th.emplace_back(worker, started);
        }
    } catch (...) { }
    for (size_t c = started; c < n_ctx; ++c) worker(c);
    for (auto &t : th) t.join();
    return TC_OK;
} TC_CATCH_STATUS(nullptr)

// ---- multiscale ICP (registration.rs:704-789) ---------------------------------------------------
tc_status tc_multiscale_icp_point_to_point(tc_context *ctx, const float *source, size_t ns, const float *target, size_t nt,
                                           const float init[7], const tc_multiscale_icp_config *cfg, tc_icp_result *result) try {
    if (!ctx || !cfg || !result || !init) return TC_INVALID_DATA;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "Source or target point cloud is empty");              // :710-714
    if (cfg->n_levels == 0) return fail(ctx, TC_INVALID_DATA, "At least one ICP scale level is required");          // :715-719
    if (!(cfg->convergence_threshold > 0.0f)) return fail(ctx, TC_INVALID_DATA, "Convergence threshold must be positive");   // :720-724
    if (cfg->final_refinement_iterations == 0) return fail(ctx, TC_INVALID_DATA, "Final refinement iterations must be positive");   // :725-729
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // full-resolution clouds and the per-level down-sampled clouds live in caller-independent buffers
    DevBuf full_s, full_t, down_s, down_t;
    auto cleanup = [&]() { for (DevBuf *b : {&full_s, &full_t, &down_s, &down_t}) if (b->p) { (void)hipFree(b->p); b->p = nullptr; } };
    tc_status st = TC_OK;
    if ((st = ensure(ctx, full_s, ns * 12)) || (st = ensure(ctx, full_t, nt * 12)) || (st = ensure(ctx, down_s, ns * 12)) ||
        (st = ensure(ctx, down_t, nt * 12))) { cleanup(); return st; }
    if (hipMemcpyAsync(full_s.p, source, ns * 12, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(full_t.p, target, nt * 12, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        cleanup();
        return fail(ctx, TC_GPU, "multiscale ICP: uploading the caller's clouds failed");
    }
    float cur[7];
    std::memcpy(cur, init, sizeof(cur));
    uint64_t total_iters = 0;
    bool any = false;
    tc_icp_result r;
    for (size_t l = 0; l < cfg->n_levels; ++l) {
        const tc_icp_scale_level &lv = cfg->levels[l];
        if (!(lv.voxel_size > 0.0f)) { cleanup(); return fail(ctx, TC_INVALID_DATA, "Scale voxel_size must be positive"); }       // :736-740
        if (lv.max_iterations == 0) { cleanup(); return fail(ctx, TC_INVALID_DATA, "Scale max_iterations must be positive"); }    // :741-745
        size_t nds = 0, ndt = 0;
        if ((st = voxel_filter_device(ctx, (const float *)full_s.p, ns, lv.voxel_size, (float *)down_s.p, &nds)) ||
            (st = voxel_filter_device(ctx, (const float *)full_t.p, nt, lv.voxel_size, (float *)down_t.p, &ndt))) { cleanup(); return st; }
        if (nds < 3 || ndt < 3) continue;                                                                             // :749-751
        std::memset(&r, 0, sizeof(r));
        st = icp_run(ctx, false, (const float *)down_s.p, nds, (const float *)down_t.p, ndt, nullptr, 0, cur, lv.max_iterations,
                     lv.max_correspondence_distance, cfg->convergence_threshold, &r, true);
        if (st != TC_OK) { cleanup(); return st; }
        std::memcpy(cur, r.transformation, sizeof(cur));
        total_iters += r.iterations;
        any = true;
    }
    if (!any) { cleanup(); return fail(ctx, TC_ALGORITHM, "No multiscale ICP level had enough downsampled points"); }   // :767-771
    tc_icp_result fin;
    std::memset(&fin, 0, sizeof(fin));
    DevBuf dcorr;
    if (result->corr_target) {
        if ((st = ensure(ctx, dcorr, ns * 4))) { cleanup(); return st; }
        fin.corr_target = (uint32_t *)dcorr.p;
    }
    st = icp_run(ctx, false, (const float *)full_s.p, ns, (const float *)full_t.p, nt, nullptr, 0, cur, cfg->final_refinement_iterations,
                 cfg->final_max_correspondence_distance, cfg->convergence_threshold, &fin, true);
    if (st == TC_OK) {
        std::memcpy(result->transformation, fin.transformation, sizeof(fin.transformation));
        result->mse = fin.mse;
        result->iterations = total_iters + fin.iterations;                                                           // :782-788
        result->converged = fin.converged;
        result->n_correspondences = fin.n_correspondences;
        if (result->corr_target) (void)hipMemcpy(result->corr_target, dcorr.p, ns * 4, hipMemcpyDeviceToHost);
    }
    if (dcorr.p) (void)hipFree(dcorr.p);
    cleanup();
    return st;
} TC_CATCH_STATUS(ctx)

// ---- KISS-ICP (kiss_icp.rs:183-300) ----------------------------------------------------------------
// range filter -> voxel down-sampling of the source -> point-to-point ICP against the full target with the
// adaptive correspondence threshold, mse measured after every update, fixed 1e-6 convergence rule
static float kiss_adaptive_threshold(const float init[7], float voxel_size) {        // :82-95, f32 like the reference
    const float trans = std::sqrt(init[4] * init[4] + init[5] * init[5] + init[6] * init[6]);
    const float imag = std::sqrt(init[0] * init[0] + init[1] * init[1] + init[2] * init[2]);
    const float motion = trans + 2.0f * imag * voxel_size;
    return std::fmin(std::fmax(3.0f * motion, 3.0f * voxel_size), 10.0f * voxel_size);
}

tc_status tc_kiss_icp_device(tc_context *ctx, const float *d_source, size_t ns, const float *d_target, size_t nt, const float init[7],
                             const tc_kiss_icp_config *cfg, tc_icp_result *result, size_t *n_source_down) try {
    if (!ctx || !cfg || !result || !init) return TC_INVALID_DATA;
    if (n_source_down) *n_source_down = 0;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "KISS-ICP: source or target point cloud is empty");     // :189-193
    if (cfg->max_iterations == 0) return fail(ctx, TC_INVALID_DATA, "KISS-ICP: max_iterations must be > 0");          // :194-198
    if (!(cfg->voxel_size > 0.0f)) return fail(ctx, TC_INVALID_DATA, "KISS-ICP: voxel_size must be > 0");             // :199-203
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf ranged, down;
    auto cleanup = [&]() { for (DevBuf *b : {&ranged, &down}) if (b->p) { (void)hipFree(b->p); b->p = nullptr; } };
    tc_status st = TC_OK;
    if ((st = ensure(ctx, ranged, ns * 12)) || (st = ensure(ctx, down, ns * 12))) { cleanup(); return st; }
    size_t nr = 0, nd = 0;
    if ((st = range_filter_device(ctx, d_source, ns, cfg->min_range, cfg->max_range, (float *)ranged.p, &nr))) { cleanup(); return st; }
    if (nr == 0) { cleanup(); return fail(ctx, TC_INVALID_DATA, "KISS-ICP: no source points remain after range filtering"); }   // :207-213
    if ((st = voxel_filter_device(ctx, (const float *)ranged.p, nr, cfg->voxel_size, (float *)down.p, &nd))) { cleanup(); return st; }
    if (nd == 0) { cleanup(); return fail(ctx, TC_INVALID_DATA, "KISS-ICP: no source points remain after voxel downsampling"); }
    if (n_source_down) *n_source_down = nd;
    const float sigma = kiss_adaptive_threshold(init, cfg->voxel_size);
    st = icp_run(ctx, false, (const float *)down.p, nd, d_target, nt, nullptr, 0, init, cfg->max_iterations, sigma, 1e-6f, result, true, 1);
    cleanup();
    return st;
} TC_CATCH_STATUS(ctx)

tc_status tc_kiss_icp(tc_context *ctx, const float *source, size_t ns, const float *target, size_t nt, const float init[7],
                      const tc_kiss_icp_config *cfg, tc_icp_result *result, size_t *n_source_down) try {
    if (!ctx || !cfg || !result || !init) return TC_INVALID_DATA;
    if (n_source_down) *n_source_down = 0;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "KISS-ICP: source or target point cloud is empty");
    if (cfg->max_iterations == 0) return fail(ctx, TC_INVALID_DATA, "KISS-ICP: max_iterations must be > 0");
    if (!(cfg->voxel_size > 0.0f)) return fail(ctx, TC_INVALID_DATA, "KISS-ICP: voxel_size must be > 0");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = ensure(ctx, ctx->in_a, ns * 12)) return s;
    if (tc_status s = ensure(ctx, ctx->in_b, nt * 12)) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_a.p, source, ns * 12, hipMemcpyHostToDevice, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_b.p, target, nt * 12, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *host_corr = result->corr_target;
    DevBuf dcorr;
    if (host_corr) {
        if (tc_status s = ensure(ctx, dcorr, ns * 4)) return s;
        result->corr_target = (uint32_t *)dcorr.p;
    }
    size_t nd = 0;
    tc_status st = tc_kiss_icp_device(ctx, (const float *)ctx->in_a.p, ns, (const float *)ctx->in_b.p, nt, init, cfg, result, &nd);
    result->corr_target = host_corr;
    if (st == TC_OK && host_corr) (void)hipMemcpy(host_corr, dcorr.p, nd * 4, hipMemcpyDeviceToHost);
    if (dcorr.p) (void)hipFree(dcorr.p);
    if (n_source_down) *n_source_down = nd;
    return st;
} TC_CATCH_STATUS(ctx)

// ---- GICP (gicp.rs:100-305) ----------------------------------------------------------------------
// compute_covariances (gicp.rs:52-86): the k nearest points INCLUDING the point itself (ascending distance),
// f32 mean and outer products in that order, / max(n - 1, 1), + 1e-4 I; fewer than 3 neighbours -> 1e-3 I.
// out: two float4 per point (xx, xy, xz, yy), (yz, zz, 0, 0), original order.
__global__ void __launch_bounds__(256) gicp_cov_kernel(const float *__restrict__ xyz, uint32_t n, const uint32_t *__restrict__ idx,
                                                      const uint32_t *__restrict__ count, uint32_t k, float4 *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = count[i];
    if (m < 3) {
        out[2 * (size_t)i] = make_float4(1e-3f, 0.f, 0.f, 1e-3f);
        out[2 * (size_t)i + 1] = make_float4(0.f, 1e-3f, 0.f, 0.f);
        return;
    }
    const uint32_t *nb = idx + (size_t)i * k;
    const float nf = (float)m;
    float mx = 0.f, my = 0.f, mz = 0.f;
    for (uint32_t j = 0; j < m; ++j) { const uint32_t q = nb[j]; mx = mx + xyz[3 * (size_t)q]; my = my + xyz[3 * (size_t)q + 1]; mz = mz + xyz[3 * (size_t)q + 2]; }
    mx /= nf; my /= nf; mz /= nf;
    float xx = 0.f, xy = 0.f, xz = 0.f, yy = 0.f, yz = 0.f, zz = 0.f;
    for (uint32_t j = 0; j < m; ++j) {
        const uint32_t q = nb[j];
        const float dx = xyz[3 * (size_t)q] - mx, dy = xyz[3 * (size_t)q + 1] - my, dz = xyz[3 * (size_t)q + 2] - mz;
        xx += dx * dx; xy += dx * dy; xz += dx * dz; yy += dy * dy; yz += dy * dz; zz += dz * dz;
    }
    const float den = fmaxf(nf - 1.0f, 1.0f);
    out[2 * (size_t)i] = make_float4(xx / den + 1e-4f, xy / den, xz / den, yy / den + 1e-4f);
    out[2 * (size_t)i + 1] = make_float4(yz / den, zz / den + 1e-4f, 0.f, 0.f);
}

static tc_status gicp_covariances_device(tc_context *ctx, const float *d_xyz, size_t n, size_t k, DevBuf &idx, DevBuf &dist, DevBuf &cnt,
                                         float *d_cov8) {
    k = std::max<size_t>(k, 4);
    if (k > 2048) return fail(ctx, TC_UNSUPPORTED, "GICP: k_correspondences > 2048 is not supported by this backend");
    if (tc_status s = ensure(ctx, idx, n * k * sizeof(uint32_t))) return s;
    if (tc_status s = ensure(ctx, dist, n * k * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, cnt, n * sizeof(uint32_t))) return s;
    // same grid as tc_knn (the point itself is one of its k nearest)
    if (tc_status s = build_index(ctx, ctx->tgt_index, d_xyz, n, normals_cell_factor(k > 1 ? k - 1 : 1, false) * 2.0f, nullptr, nullptr)) return s;
    if (tc_status s = launch_knn(ctx, ctx->tgt_index, d_xyz, n, k, (uint32_t *)idx.p, (float *)dist.p, (uint32_t *)cnt.p)) return s;
    ProfScope ps(ctx, "gicp_covariances");
    hipLaunchKernelGGL(gicp_cov_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_xyz, (uint32_t)n, (const uint32_t *)idx.p,
                       (const uint32_t *)cnt.p, (uint32_t)k, (float4 *)d_cov8);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

tc_status tc_gicp_device(tc_context *ctx, const float *d_source, size_t ns, const float *d_target, size_t nt, const float init[7],
                         const tc_gicp_config *cfg, tc_icp_result *result) try {
    if (!ctx || !cfg || !result || !init) return TC_INVALID_DATA;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "GICP: source or target point cloud is empty");           // :107-111
    if (cfg->max_iterations == 0) return fail(ctx, TC_INVALID_DATA, "GICP: max_iterations must be > 0");                // :112-116
    const size_t min_k = std::max<size_t>(cfg->k_correspondences, 4);
    if (ns < min_k || nt < min_k) return fail(ctx, TC_INVALID_DATA, "GICP: clouds must have at least k_correspondences points");   // :120-131
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float *clouds[2] = {d_source, d_target};
    const size_t sizes[2] = {ns, nt};
    for (int c = 0; c < 2; ++c) {                                                                                       // :135-155
        float mn[3], mx[3];
        if (tc_status s = cloud_bbox(ctx, clouds[c], sizes[c], mn, mx)) return s;
        const float me = std::fmin(std::fmin(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
        if (me < 1e-4f) return fail(ctx, TC_INVALID_DATA, "GICP: point cloud appears to be coplanar or collinear; GICP requires 3-D structure");
    }
    DevBuf idx, dist, cnt, cov_s, cov_t;
    auto cleanup = [&]() { for (DevBuf *b : {&idx, &dist, &cnt, &cov_s, &cov_t}) if (b->p) { (void)hipFree(b->p); b->p = nullptr; } };
    tc_status st = TC_OK;
    if ((st = ensure(ctx, cov_s, ns * 8 * sizeof(float))) || (st = ensure(ctx, cov_t, nt * 8 * sizeof(float))) ||
        (st = gicp_covariances_device(ctx, d_source, ns, cfg->k_correspondences, idx, dist, cnt, (float *)cov_s.p)) ||
        (st = gicp_covariances_device(ctx, d_target, nt, cfg->k_correspondences, idx, dist, cnt, (float *)cov_t.p))) { cleanup(); return st; }
    st = icp_run_gicp(ctx, d_source, ns, d_target, nt, (const float *)cov_s.p, (const float *)cov_t.p, init, cfg->max_iterations,
                      cfg->max_correspondence_distance, cfg->convergence_threshold, result, true);
    cleanup();
    return st;
} TC_CATCH_STATUS(ctx)

tc_status tc_gicp(tc_context *ctx, const float *source, size_t ns, const float *target, size_t nt, const float init[7],
                  const tc_gicp_config *cfg, tc_icp_result *result) try {
    if (!ctx || !cfg || !result || !init) return TC_INVALID_DATA;
    if (ns == 0 || nt == 0) return fail(ctx, TC_INVALID_DATA, "GICP: source or target point cloud is empty");
    if (cfg->max_iterations == 0) return fail(ctx, TC_INVALID_DATA, "GICP: max_iterations must be > 0");
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = ensure(ctx, ctx->in_a, ns * 12)) return s;
    if (tc_status s = ensure(ctx, ctx->in_b, nt * 12)) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_a.p, source, ns * 12, hipMemcpyHostToDevice, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_b.p, target, nt * 12, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *host_corr = result->corr_target;
    DevBuf dcorr;
    if (host_corr) {
        if (tc_status s = ensure(ctx, dcorr, ns * 4)) return s;
        result->corr_target = (uint32_t *)dcorr.p;
    }
    tc_status st = tc_gicp_device(ctx, (const float *)ctx->in_a.p, ns, (const float *)ctx->in_b.p, nt, init, cfg, result);
    result->corr_target = host_corr;
    if (st == TC_OK && host_corr) (void)hipMemcpy(host_corr, dcorr.p, ns * 4, hipMemcpyDeviceToHost);
    if (dcorr.p) (void)hipFree(dcorr.p);
    return st;
} TC_CATCH_STATUS(ctx)

// ---- batch k-NN (nearest_neighbor.rs:177-251; gpu/nearest_neighbor.rs:332-355) ----------------
tc_status tc_knn_device(tc_context *ctx, const float *d_cloud, size_t n, const float *d_queries, size_t nq, size_t k,
                        uint32_t *d_idx, float *d_dist, uint32_t *d_count) try {
    if (!ctx) return TC_INVALID_DATA;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (k == 0 || n == 0) {        // nearest_neighbor.rs:178-180: empty result
        TC_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, nq * sizeof(uint32_t), ctx->stream));
        TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return TC_OK;
    }
    if (n >= 0xFFFFFFF0ull || nq >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    if (k > 2048) return fail(ctx, TC_UNSUPPORTED, "k > 2048 is not supported by the HIP k-NN export");
    if (tc_status s = build_index(ctx, ctx->tgt_index, d_cloud, n, normals_cell_factor(k > 1 ? k - 1 : 1, false) * 2.0f, nullptr, nullptr)) return s;
    if (tc_status s = launch_knn(ctx, ctx->tgt_index, d_queries, nq, k, d_idx, d_dist, d_count)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

// ---- radius search export (nearest_neighbor.rs:254-298; gpu_find_radius_neighbors gpu/nearest_neighbor.rs:357-367) ----
tc_status tc_radius_search_device(tc_context *ctx, const float *d_cloud, size_t n, const float *d_queries, size_t nq, float radius, size_t k_max,
                                  uint32_t *d_idx, float *d_dist, uint32_t *d_count) try {
    if (!ctx) return TC_INVALID_DATA;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!(radius > 0.0f) || n == 0 || k_max == 0) {      // nearest_neighbor.rs:255-257: empty result
        TC_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, nq * sizeof(uint32_t), ctx->stream));
        TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return TC_OK;
    }
    if (n >= 0xFFFFFFF0ull || nq >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    if (k_max > 2048) return fail(ctx, TC_UNSUPPORTED, "k_max > 2048 is not supported by the HIP radius search");
    if (tc_status s = build_index(ctx, ctx->tgt_index, d_cloud, n, normals_cell_factor(k_max > 1 ? k_max - 1 : 1, false) * 2.0f, nullptr, nullptr)) return s;
    if (tc_status s = launch_knn(ctx, ctx->tgt_index, d_queries, nq, k_max, d_idx, d_dist, d_count, radius * radius)) return s;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_radius_search(tc_context *ctx, const float *cloud, size_t n, const float *queries, size_t nq, float radius, size_t k_max,
                           uint32_t *idx, float *dist, uint32_t *count) try {
    if (!ctx) return TC_INVALID_DATA;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!(radius > 0.0f) || n == 0 || k_max == 0) { std::memset(count, 0, nq * sizeof(uint32_t)); return TC_OK; }
    if (k_max > 2048) return fail(ctx, TC_UNSUPPORTED, "k_max > 2048 is not supported by the HIP radius search");
    DevBuf dc, dq, di, dd, dn;
    auto cleanup = [&]() { for (DevBuf *b : {&dc, &dq, &di, &dd, &dn}) if (b->p) { (void)hipFree(b->p); b->p = nullptr; } };
    tc_status st = TC_OK;
    if ((st = ensure(ctx, dc, n * 12)) || (st = ensure(ctx, dq, nq * 12)) || (st = ensure(ctx, di, nq * k_max * 4)) ||
        (st = ensure(ctx, dd, nq * k_max * 4)) || (st = ensure(ctx, dn, nq * 4))) { cleanup(); return st; }
    (void)hipMemcpyAsync(dc.p, cloud, n * 12, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(dq.p, queries, nq * 12, hipMemcpyHostToDevice, ctx->stream);
    st = tc_radius_search_device(ctx, (const float *)dc.p, n, (const float *)dq.p, nq, radius, k_max, (uint32_t *)di.p, (float *)dd.p, (uint32_t *)dn.p);
    if (st == TC_OK) {
        (void)hipMemcpy(idx, di.p, nq * k_max * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(dist, dd.p, nq * k_max * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(count, dn.p, nq * 4, hipMemcpyDeviceToHost);
    }
    cleanup();
    return st;
} TC_CATCH_STATUS(ctx)

tc_status tc_knn(tc_context *ctx, const float *cloud, size_t n, const float *queries, size_t nq, size_t k,
                 uint32_t *idx, float *dist, uint32_t *count) try {
    if (!ctx) return TC_INVALID_DATA;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (k == 0 || n == 0) { std::memset(count, 0, nq * sizeof(uint32_t)); return TC_OK; }
    if (k > 2048) return fail(ctx, TC_UNSUPPORTED, "k > 2048 is not supported by the HIP k-NN export");      // before any buffer is sized by k
    if (tc_status s = ensure(ctx, ctx->in_a, n * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->in_b, nq * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->out_a, nq * k * 8 + nq * 4)) return s;
    uint32_t *d_idx = (uint32_t *)ctx->out_a.p;
    float *d_dist = (float *)(d_idx + nq * k);
    uint32_t *d_cnt = (uint32_t *)(d_dist + nq * k);
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_a.p, cloud, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_b.p, queries, nq * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status s = tc_knn_device(ctx, (const float *)ctx->in_a.p, n, (const float *)ctx->in_b.p, nq, k, d_idx, d_dist, d_cnt)) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(idx, d_idx, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(dist, d_dist, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(count, d_cnt, nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

// ---- persistent search index: KdTree::new once, many find_k_nearest / find_radius_neighbors calls --------------
// (threecrate-core/src/traits.rs:6-12; nearest_neighbor.rs:37-58, :177-298; Python KdTree lib.rs:707-776)
}  // extern "C" (reopened below)

struct tc_search_index {
    tc_context *ctx;
    tc::DeviceIndex ix;
    size_t n;
    tc::DevBuf q, out;          // staged queries / results of the host-buffer calls
};

extern "C" {

tc_status tc_search_index_create_device(tc_context *ctx, const float *d_cloud, size_t n, size_t k_hint, tc_search_index **out) try {
    if (!ctx || !out) return TC_INVALID_DATA;
    *out = nullptr;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    tc_search_index *s = new tc_search_index{ctx, {}, n, {}, {}};
    if (n) {        // an empty cloud is an empty tree (nearest_neighbor.rs:38-45)
        const size_t k = std::min<size_t>(std::max<size_t>(k_hint, 1), 129);
        tc_status rc = build_index(ctx, s->ix, d_cloud, n, normals_cell_factor(k > 1 ? k - 1 : 1, false) * 2.0f, nullptr, nullptr);
        if (rc == TC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, TC_GPU, "search index build failed");
        if (rc != TC_OK) { free_index(s->ix); delete s; return rc; }
        // queries only need the sorted records and the cell starts: drop the build scratch (16 B per point)
        free_buf(s->ix.cell_of); free_buf(s->ix.slot); free_buf(s->ix.arrival); free_buf(s->ix.fill); free_buf(s->ix.blocksum);
    }
    *out = s;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

tc_status tc_search_index_create(tc_context *ctx, const float *cloud, size_t n, size_t k_hint, tc_search_index **out) try {
    if (!ctx || !out) return TC_INVALID_DATA;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n) {
        if (tc_status s = ensure(ctx, ctx->in_a, n * 3 * sizeof(float))) return s;
        TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_a.p, cloud, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    return tc_search_index_create_device(ctx, (const float *)ctx->in_a.p, n, k_hint, out);   // the index holds its own sorted copy
} TC_CATCH_STATUS(ctx)

size_t tc_search_index_size(const tc_search_index *s) { return s ? s->n : 0; }

// radius < 0: k nearest; radius >= 0: the neighbours within radius among the k nearest
tc_status tc_search_index_query_device(tc_search_index *s, const float *d_queries, size_t nq, size_t k, float radius,
                                       uint32_t *d_idx, float *d_dist, uint32_t *d_count) try {
    if (!s) return TC_INVALID_DATA;
    tc_context *ctx = s->ctx;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool by_radius = radius >= 0.0f;
    if (k == 0 || s->n == 0 || (by_radius && !(radius > 0.0f))) {        // nearest_neighbor.rs:178-180, :255-257
        TC_HIP_TRY(ctx, hipMemsetAsync(d_count, 0, nq * sizeof(uint32_t), ctx->stream));
        TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return TC_OK;
    }
    if (k > 2048) return fail(ctx, TC_UNSUPPORTED, "k > 2048 is not supported by the HIP neighbour search");
    if (nq >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    if (tc_status rc = launch_knn(ctx, s->ix, d_queries, nq, k, d_idx, d_dist, d_count, by_radius ? radius * radius : INFINITY)) return rc;
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_search_index_query(tc_search_index *s, const float *queries, size_t nq, size_t k, float radius, uint32_t *idx, float *dist,
                                uint32_t *count) try {
    if (!s) return TC_INVALID_DATA;
    tc_context *ctx = s->ctx;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (k == 0 || s->n == 0) { std::memset(count, 0, nq * sizeof(uint32_t)); return TC_OK; }
    if (k > 2048) return fail(ctx, TC_UNSUPPORTED, "k > 2048 is not supported by the HIP neighbour search");
    if (tc_status rc = ensure(ctx, s->q, nq * 3 * sizeof(float))) return rc;
    if (tc_status rc = ensure(ctx, s->out, nq * k * 8 + nq * 4)) return rc;
    uint32_t *d_idx = (uint32_t *)s->out.p;
    float *d_dist = (float *)(d_idx + nq * k);
    uint32_t *d_cnt = (uint32_t *)(d_dist + nq * k);
    TC_HIP_TRY(ctx, hipMemcpyAsync(s->q.p, queries, nq * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status rc = tc_search_index_query_device(s, (const float *)s->q.p, nq, k, radius, d_idx, d_dist, d_cnt)) return rc;
    TC_HIP_TRY(ctx, hipMemcpyAsync(idx, d_idx, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(dist, d_dist, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(count, d_cnt, nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

// find_radius_neighbors without a cap (nearest_neighbor.rs:254-298): count, then fill at the caller's offsets
tc_status tc_search_index_radius_count(tc_search_index *s, const float *queries, size_t nq, float radius, uint32_t *counts) try {
    if (!s) return TC_INVALID_DATA;
    tc_context *ctx = s->ctx;
    if (nq == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!(radius > 0.0f) || s->n == 0) { std::memset(counts, 0, nq * sizeof(uint32_t)); return TC_OK; }      // :255-257
    if (nq >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    if (tc_status rc = ensure(ctx, s->q, nq * 3 * sizeof(float))) return rc;
    if (tc_status rc = ensure(ctx, s->out, nq * sizeof(uint32_t))) return rc;
    TC_HIP_TRY(ctx, hipMemcpyAsync(s->q.p, queries, nq * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status rc = launch_radius_all(ctx, s->ix, (const float *)s->q.p, nq, radius, (uint32_t *)s->out.p, nullptr, nullptr, nullptr)) return rc;
    TC_HIP_TRY(ctx, hipMemcpyAsync(counts, s->out.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_search_index_radius_fill(tc_search_index *s, const float *queries, size_t nq, float radius, const uint64_t *offsets, size_t total,
                                      uint32_t *idx, float *dist) try {
    if (!s) return TC_INVALID_DATA;
    tc_context *ctx = s->ctx;
    if (nq == 0 || total == 0) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!(radius > 0.0f) || s->n == 0) return fail(ctx, TC_INVALID_DATA, "radius fill: nothing to fill for this radius (total must be 0)");
    const size_t q_bytes = (nq * 3 * sizeof(float) + 7) / 8 * 8;
    if (tc_status rc = ensure(ctx, s->q, q_bytes + nq * sizeof(uint64_t))) return rc;
    if (tc_status rc = ensure(ctx, s->out, total * 8)) return rc;
    float *d_q = (float *)s->q.p;
    unsigned long long *d_off = (unsigned long long *)((char *)s->q.p + q_bytes);
    uint32_t *d_idx = (uint32_t *)s->out.p;
    float *d_dist = (float *)(d_idx + total);
    TC_HIP_TRY(ctx, hipMemcpyAsync(d_q, queries, nq * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(d_off, offsets, nq * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status rc = launch_radius_all(ctx, s->ix, d_q, nq, radius, nullptr, d_off, d_idx, d_dist)) return rc;
    TC_HIP_TRY(ctx, hipMemcpyAsync(idx, d_idx, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipMemcpyAsync(dist, d_dist, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

void tc_search_index_destroy(tc_search_index *s) try {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    free_index(s->ix);
    free_buf(s->q); free_buf(s->out);
    delete s;
} TC_CATCH_VOID

// ---- voxel_grid_filter (filtering.rs:38-133) --------------------------------------------------
static tc_status voxel_validate(tc_context *ctx, size_t n, float voxel, size_t *n_out, bool *empty) {
    *empty = false;
    if (!ctx || !n_out) return TC_INVALID_DATA;
    *n_out = 0;
    if (n == 0) { *empty = true; return TC_OK; }                                              // filtering.rs:42-44
    if (!(voxel > 0.0f)) return fail(ctx, TC_INVALID_DATA, "voxel_size must be positive");    // :46-50
    if (n >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    return TC_OK;
}

tc_status tc_voxel_grid_filter_device(tc_context *ctx, const float *d_xyz, size_t n, float voxel_size, float *d_out, size_t *n_out) try {
    bool empty;
    if (tc_status s = voxel_validate(ctx, n, voxel_size, n_out, &empty)) return s;
    if (empty) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return voxel_filter_device(ctx, d_xyz, n, voxel_size, d_out, n_out);
} TC_CATCH_STATUS(ctx)

tc_status tc_voxel_grid_filter(tc_context *ctx, const float *xyz, size_t n, float voxel_size, float *out, size_t *n_out) try {
    bool empty;
    if (tc_status s = voxel_validate(ctx, n, voxel_size, n_out, &empty)) return s;
    if (empty) return TC_OK;
    TC_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (tc_status s = ensure(ctx, ctx->in_a, n * 3 * sizeof(float))) return s;
    if (tc_status s = ensure(ctx, ctx->out_a, n * 3 * sizeof(float))) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->in_a.p, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (tc_status s = voxel_filter_device(ctx, (const float *)ctx->in_a.p, n, voxel_size, (float *)ctx->out_a.p, n_out)) return s;
    TC_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->out_a.p, *n_out * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS(ctx)

// ---- profiling ------------------------------------------------------------------------------
void tc_profile_enable(tc_context *ctx, int on) {
    if (!ctx) return;
    ctx->profiling = on;
    ctx->prof_tick = 0;
    if (on == 3) for (auto &v : ctx->stat_icp) v = 0;          // a statistics session starts from zero
}

static void profile_collect(tc_context *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->timers) {
        for (auto &p : t.pending) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                t.total_ms += ms; t.launches += 1;
                t.min_ms = std::min(t.min_ms, (double)ms); t.max_ms = std::max(t.max_ms, (double)ms);
            }
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
        t.pending.clear();
    }
}

void tc_profile_reset(tc_context *ctx) try {
    if (!ctx) return;
    profile_collect(ctx);
    for (auto &t : ctx->timers) { t.launches = 0; t.total_ms = 0.0; t.min_ms = 1e300; t.max_ms = 0.0; }
} TC_CATCH_VOID

size_t tc_profile_read(tc_context *ctx, tc_kernel_stat *out, size_t cap) try {
    if (!ctx) return 0;
    profile_collect(ctx);
    size_t n = 0;
    for (auto &t : ctx->timers) {
        if (n < cap && out) {
            std::memset(&out[n], 0, sizeof(out[n]));
            std::strncpy(out[n].name, t.name.c_str(), sizeof(out[n].name) - 1);
            out[n].launches = t.launches;
            out[n].total_ms = t.total_ms;
            out[n].min_ms = t.launches ? t.min_ms : 0.0;
            out[n].max_ms = t.max_ms;
        }
        ++n;
    }
    return n;
} TC_CATCH_VALUE(0)

}  // extern "C"
