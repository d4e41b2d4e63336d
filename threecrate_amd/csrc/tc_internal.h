// tc_internal.h -- shared host/device declarations of libthreecrate_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <new>
#include <exception>
#include <string>
#include <vector>

#include "../../include/threecrate_hip.h"

namespace tc {

// ------------------------------------------------------------------------------------------
// Uniform-grid spatial index over one cloud (replaces KdTree, nearest_neighbor.rs:29-33).
// Cells are ordered x-fastest, so a run of cells along x is ONE contiguous range of the
// cell-sorted point array: a (2R+1)^3 neighbourhood is (2R+1)^2 contiguous spans.
// ------------------------------------------------------------------------------------------
struct GridGeom {
    float minx, miny, minz;     // bbox min (cell origin)
    float maxx, maxy, maxz;     // bbox max
    float h, inv_h;             // cell edge and 1/h
    int   gx, gy, gz;           // cells per axis
    uint32_t ncell;             // gx*gy*gz
    uint32_t n;                 // points indexed
    float cx, cy, cz;           // bbox centre (shift origin for the p2p Kabsch sums)
    int   clamped;              // the box is narrower than the cloud (far outliers): points beyond it live in the
                                // boundary cells, which then extend to infinity for every distance bound
};

// Tiles of TX x TY x TZ cells.  A workgroup owns one tile of QUERIES and stages the tile +
// halo region of the cell-sorted target records into LDS.  The ICP source cloud is sorted by
// the tile-major id (tile index * cells_per_tile + x-fastest local cell) so that the queries of
// one tile are one contiguous range.
struct TileGeom {
    int tx, ty, tz;             // cells per tile along each axis
    int ntx, nty, ntz;          // tiles per axis (grid dims rounded up)
    uint32_t cpt;               // cells per tile
    uint32_t ntiles;
};

struct GridView {
    GridGeom g;
    const float4   *pts;        // cell-sorted points: x, y, z, w = bit pattern of the original index
    const uint32_t *cell_start; // ncell + 1 exclusive prefix sums
    const float    *pts12;      // the same records as packed 12-byte x, y, z (only once the index has served as an ICP target; else null)
};

// device-side ICP state (one per running registration); mirrors the loop variables of
// registration.rs:278-340 / :533-593.
struct IcpState {
    float    q[4];          // current_transform rotation (i j k w)
    float    t[3];          // current_transform translation
    float    prev_mse;      // previous_mse (starts +inf)
    float    mse;           // mse of the last executed iteration
    uint32_t iterations;    // executed iterations
    int32_t  converged;
    int32_t  status;        // tc_status
    int32_t  done;          // converged or failed: later launches exit immediately
    uint32_t n_corr;        // valid pairs of the last executed iteration
    float    conv_thr;
    float    max_dist;      // < 0 : none
    int32_t  kiss;          // KISS-ICP rules (kiss_icp.rs): mse after the update, |H| check, last mse when not converged
    uint32_t refine_total;  // statistics: queries served by the refine pass (sum / max over iterations)
    uint32_t refine_max;
    uint32_t refine_ring_hist[8];   // TC_REFINE_STATS builds only: exit ring of the refine queries
    float    d_ang;         // the last update moved a point x by at most d_ang |x| + d_t (2 |sin(theta / 2)| and |translation| of the delta);
    float    d_t;           // d_ang < 0: the second-neighbour certificate is off (the update is still large: icp.hip compose())
    uint32_t d_run;         // consecutive updates small enough for it: the main pass maintains the bounds from 1 on and USES them from 2 on
    uint32_t searchers;     // lanes of the last main pass that had to search (its rows' spare column, summed by icp_finalize): the certificate's gate
    double   sums[TC_ICP_SUMS_STRIDE];   // packed, fully reduced sums of the current iteration
};

static_assert(offsetof(IcpState, refine_ring_hist) % 8 == 4 && offsetof(IcpState, sums) % 8 == 0,
              "refine_ring_hist[3..4] is ONE 64-bit counter of the statistics instantiation (icp.hip): it must sit on an 8-byte boundary");
constexpr int kIcpBlock = 256;
// padding behind the sorted records / the prefix sums: the ICP search reads a few entries past a
// span (4-wide steps) and 16-byte windows of cell_start without clamping
constexpr size_t kPtsPad = 4, kCellStartPad = 4;
constexpr size_t kCellStartFront = 4;            // zero entries in front of the prefix sums (16-byte aligned start)
constexpr uint32_t kRankQuadraticMax = 1u << 20; // cells up to this population are re-ranked by original index in O(m^2) (rerank_kernel; 65536 until round 4)
constexpr int kMaxPartialBlocks = 1024;         // plan_launch: one round of 4 blocks per CU
// clouds from this size on get the occupancy-adapted cell edge (one host round trip + possibly a rebuild)
constexpr uint32_t kAdaptMinPoints = 1u << 18;   // 2^17: a 230 k-point depth frame gets slower (normals 0.67 -> 0.71 ms, 10 ICP iterations 1.5 -> 2.8 ms)

// ---- device helpers -----------------------------------------------------------------------
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t tile_major_id(const TileGeom &t, int cx, int cy, int cz) {
    const int ax = cx / t.tx, ay = cy / t.ty, az = cz / t.tz;
    const int lx = cx - ax * t.tx, ly = cy - ay * t.ty, lz = cz - az * t.tz;
    const uint32_t tile = ((uint32_t)az * t.nty + ay) * t.ntx + ax;
    return tile * t.cpt + ((uint32_t)lz * t.ty + ly) * t.tx + lx;
}
__device__ __forceinline__ float d2_nc(float ax, float ay, float az, float bx, float by, float bz) {
    // nearest_neighbor.rs:162-167: (a - b) per component, dx*dx + dy*dy + dz*dz, left to right,
    // NO fma contraction (the library is built with -ffp-contract=off).
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ int cell_coord(float v, float mn, float inv_h, int g) {
    float f = (v - mn) * inv_h;
    int c = (f >= 0.0f) ? (int)f : 0;          // NaN -> 0
    return c < g ? c : g - 1;
}
#endif

// ---- host side ----------------------------------------------------------------------------
struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
};

struct KernelTimer {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    uint64_t launches = 0;
    double   total_ms = 0.0;
    double   min_ms = 1e300, max_ms = 0.0;
};

// One indexed cloud living in device memory.
struct DeviceIndex {
    GridGeom geom{};
    TileGeom tile{};    // only meaningful for a tile-major (query-side) ordering
    float exact_min[3] = {0, 0, 0}, exact_max[3] = {0, 0, 0};   // the cloud's exact box (geom may be clamped)
    DevBuf pts;         // float4 * n   (cell-sorted, w = original index bits)
    DevBuf cell_start;  // u32 * (ncell+1)
    DevBuf normals;     // float4 * n   (cell-sorted target normals; optional)
    DevBuf vor;         // float4 * n   (ICP target: x, y, z + inscribed-ball bound, icp_target_nn_bound_kernel; optional)
    bool vor_valid = false;     // vor belongs to the current contents of pts (build_index resets it)
    DevBuf pts12;       // float * 3 (n + 16): the sorted records again as packed 12-byte x, y, z -- the ICP candidate loop reads four
                        // of them with THREE 16-byte reads (made by icp_setup when the index first serves as an ICP target)
    bool pts12_valid = false;   // (build_index resets it)
    uint32_t occ_host = 0;      // occupied cells of the final grid, when the build read them back (edge adaptation: clouds of >= 2^18 points)
    bool occ_host_valid = false;
    DevBuf cell_of;     // u32 * n      (scratch: cell id per original point)
    DevBuf slot;        // u32 * n      (scratch: atomic scatter order)
    DevBuf arrival;     // u32 * n      (scratch: arrival rank of a point inside its cell)
    DevBuf fill;        // u32 * ncell  (scratch: histogram / fill counters)
    DevBuf blocksum;    // u32 * nblocks (scan scratch)
};

}  // namespace tc

struct tc_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string last_error;
    int profiling = 0;          // 0 off, 1 every kernel, 2 only the dominant kernel, every 17th launch
    uint32_t prof_tick = 0;
    std::vector<tc::KernelTimer> timers;
    std::vector<hipEvent_t> event_pool;
    // device blocks given back by destroyed handles (tc_cloud, tc_search_index), reused by the next allocation of a similar
    // size: a handle per frame must not cost a hipMalloc / hipFree pair per buffer per frame (milliseconds)
    std::vector<tc::DevBuf> pool;
    size_t pool_bytes = 0;
    size_t pool_largest = 0;                // largest block ever parked: the pool's cap is a multiple of it (api.hip: recycle)
    hipEvent_t order_event = nullptr;       // tc_context_wait_stream
    hipEvent_t release_event = nullptr;     // tc_stream_wait_context
    std::vector<hipEvent_t> chunk_events;   // hipEventDisableTiming events of the ICP loop's chunk polling, reused across calls
    // host entry points: the caller's source (+ normals) are uploaded on a second stream while the context's stream indexes the
    // target; upload_pending = the context's stream has to wait for upload_event before it touches them (icp_setup does)
    hipStream_t copy_stream = nullptr;
    hipEvent_t upload_event = nullptr;
    bool upload_pending = false;
    bool upload_used_copy_stream = false;   // some upload of the current call went to the copy stream (small ones do not)

    // persistent (grow-only) device buffers, reused across calls
    tc::DeviceIndex tgt_index;      // target / normals cloud
    tc::DeviceIndex src_index;      // source ordered by target cell (ICP)
    tc::DevBuf in_a, in_b, in_c;    // staged host inputs
    tc::DevBuf out_a;               // staged outputs
    tc::DevBuf bbox;                // 6 x u32 (ordered-int encoded floats)
    tc::DevBuf state;               // IcpState
    tc::DevBuf partials;            // double * kMaxPartialBlocks * TC_ICP_SUMS_STRIDE
    tc::DevBuf corr;                // u32 * n_source
    tc::DevBuf gicp_src_cov;        // GICP: source covariances in the sorted source order (2 float4 per point)
    tc::DevBuf icp_wsrc;            // float4 * n_source: the ICP loop's working copy of the ordered source: x, y, z + the position of the
                                    // current match in w (one 16-byte read per point and iteration instead of record + match)
    tc::DevBuf dbg_times;           // TC_DEBUG & 1024: per-block stamps of the main pass
    tc::DevBuf overflow;            // scratch (voxel filter: occupied-cell flags / output slots)
    tc::DevBuf build_tmp;           // index build: the records in arrival order, before the in-cell re-rank (float4 * n)
    tc::DevBuf normals_hard;        // normals: count + positions of the points handed to the wave-per-point kernel
    unsigned long long stat_indexed_points = 0, stat_index_builds = 0;   // tc_debug_counter
    // search statistics of the ICP main pass, summed over the calls made in profiling mode 3 (tc_profile_enable(ctx, 3)):
    // iterations, wave trips, trips without a search, searches, candidate steps needed, candidate steps taken (slowest lanes)
    unsigned long long stat_icp[6] = {0, 0, 0, 0, 0, 0};
    bool icp_cert = false;          // run_chunked: the chunks being enqueued run the certificate's instantiation of the main pass
    bool icp_cert_hint = false;     // ... and what the context's previous registration ended with (the next one starts with it)
    bool normals_hard_clean = false; // its header (count, exit ticket) is known to be zero: the last serving launch went through
    tc::DeviceIndex vox_index;      // voxel filter counting-sort buffers
    void *pinned = nullptr;         // small pinned host scratch (IcpState readback, bbox)
    void *pinned_dev = nullptr;     // the device's address of the same block
    size_t pinned_cap = 0;
};

struct tc_comm {
    tc_context *ctx = nullptr;
    int rank = 0, nranks = 1;
    void *nccl = nullptr;               // ncclComm_t (RCCL), or null
    bool own_nccl = false;
    tc_host_collective_fn host_fn = nullptr;
    void *host_user = nullptr;
    void *agree_word = nullptr;         // one device u32: comm_agree
};

namespace tc {

// error plumbing
tc_status fail(tc_context *ctx, tc_status st, const std::string &msg);
tc_status fail_nothrow(tc_context *ctx, tc_status st, const char *msg) noexcept;
void fault_point(const char *site);       // TC_FAULT: test-only fault injection (api.hip)
// No C++ exception crosses the C ABI (threecrate-core/src/error.rs:7-28: every failure is an Error value; unwinding into a Rust or C
// caller is undefined behaviour): every extern "C" entry point with a body that can allocate is a function-try-block closed by one
// of these, and every thread body catches for itself.
#define TC_CATCH_STATUS(CTX)                                                                                            \
    catch (const std::bad_alloc &) { return tc::fail_nothrow((CTX), TC_GPU, "out of host memory"); }                    \
    catch (const std::exception &e_) { return tc::fail_nothrow((CTX), TC_GPU, e_.what()); }                             \
    catch (...) { return tc::fail_nothrow((CTX), TC_GPU, "unknown C++ exception"); }
#define TC_CATCH_VOID catch (...) { }
#define TC_CATCH_VALUE(V) catch (...) { return V; }
#define TC_HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return tc::fail((ctx), TC_GPU, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

tc_status ensure(tc_context *ctx, DevBuf &b, size_t bytes);
// host -> device on the context's copy stream (created on first use), followed by tc::uploads_issued(): the context's stream
// waits for everything issued so far the next time tc::wait_uploads is called
tc_status upload_async(tc_context *ctx, void *d_dst, const void *h_src, size_t bytes);
tc_status uploads_issued(tc_context *ctx);
tc_status wait_uploads(tc_context *ctx);
// Device -> host words without a copy kernel or a stream synchronisation: a kernel stores into the context's pinned (host-coherent)
// block through its device address and raises a flag word there last (system scope); the host spins on the flag -- with a
// hipStreamQuery now and then, never on a dead device.  dev_ptr: the device's view of a host address inside ctx->pinned.
void *pinned_dev_ptr(tc_context *ctx, const void *host_addr);
bool pinned_poll_enabled();       // TC_NO_PINNED_POLL=1: copies + stream synchronisations as before round 4 (A/B)
tc_status wait_pinned_word(tc_context *ctx, volatile uint32_t *word, const char *what);
// hand a block back to the context's pool (the caller has made sure no work in flight uses it)
void recycle(tc_context *ctx, DevBuf &b);

// profiling scope: records hipEvents around one kernel launch on ctx->stream.
// A `dominant` scope (the ICP main pass: the kernel of the bench line's roofline) hands its two events to the launch itself
// (hipExtLaunchKernelGGL(.., e0, e1, ..): they take the kernel's own start and end stamps, no marker packets on the stream -- what
// rocprofv3's kernel trace reads; events recorded AROUND the launch measured 2.3 us more than the kernel ran and put a bubble on
// either side of it): the launch site tests active() and launches through launch_timed().
struct ProfScope {
    tc_context *ctx; int idx = -1; hipEvent_t e0 = nullptr, e1 = nullptr; bool ext = false;
    ProfScope(tc_context *c, const char *name, bool dominant = false);
    ~ProfScope();
    bool active() const { return idx >= 0; }
};

// grid.hip
tc_status build_index(tc_context *ctx, DeviceIndex &ix, const float *d_xyz, size_t n,
                      float cell_factor, const GridGeom *reuse_geom, const IcpState *d_state_transform,
                      const TileGeom *tile_major = nullptr, float min_cell_edge = 0.0f, float target_ppo = 0.0f, bool strict_order = false);
TileGeom make_tiles(const GridGeom &g, int tx, int ty, int tz);
tc_status gather_normals(tc_context *ctx, DeviceIndex &ix, const float *d_normals, size_t stride);
GridView view_of(const DeviceIndex &ix);

// grid.hip (shared with voxel.hip)
tc_status exclusive_scan_u32(tc_context *ctx, const uint32_t *d_in, uint32_t n, uint32_t *d_out /* n+1 */, DevBuf &blocksum, uint32_t *occ_out = nullptr,
                             unsigned long long *occ_host = nullptr);
tc_status cloud_bbox(tc_context *ctx, const float *d_xyz, size_t n, float mn[3], float mx[3]);

// voxel.hip
tc_status voxel_filter_device(tc_context *ctx, const float *d_xyz, size_t n, float voxel, float *d_out, size_t *n_out);
tc_status range_filter_device(tc_context *ctx, const float *d_xyz, size_t n, float min_range, float max_range, float *d_out, size_t *n_out);

// normals.hip
tc_status launch_normals(tc_context *ctx, const DeviceIndex &ix, const float *d_xyz, const tc_normal_config &cfg,
                         const float vp[3], float *d_out6, size_t p_begin = 0, size_t p_end = (size_t)-1, bool slice_out = false,
                         float4 *d_sorted_nrm = nullptr, float4 *d_vor = nullptr);
// api.hip: index (into `ix`) + normals of a device-resident cloud
tc_status normals_on_index(tc_context *ctx, DeviceIndex &ix, bool build, float cell_factor_override, const float *d_xyz, size_t n,
                           const tc_normal_config *cfg, float *d_out6, size_t p_begin, size_t p_end, bool slice_out, float4 *d_sorted_nrm,
                           bool with_bounds = false);
float normals_cell_factor(size_t k, bool large);
float normals_target_ppo(size_t k);
float icp_cell_factor();
void free_index(DeviceIndex &ix);
void recycle_index(tc_context *ctx, DeviceIndex &ix);       // blocks back to the context's pool
tc_status launch_radius_all(tc_context *ctx, const DeviceIndex &ix, const float *d_queries, size_t nq, float radius, uint32_t *d_counts,
                            const unsigned long long *d_offsets, uint32_t *d_idx, float *d_dist);
tc_status launch_normals_unsort(tc_context *ctx, const DeviceIndex &ix, const float *d_sorted6, float *d_out6);

tc_status launch_knn(tc_context *ctx, const DeviceIndex &ix, const float *d_queries, size_t nq, size_t k,
                     uint32_t *d_idx, float *d_dist, uint32_t *d_count, float radius_sq = INFINITY);

// TC_DEBUG bit mask, read from the environment once per process
int debug_flags();

// comm.hip: collectives of a tc_comm on the context's stream (RCCL) or through the host callback (blocking)
tc_status comm_allreduce_f64(tc_comm *comm, double *d_buf, size_t count);
tc_status comm_allreduce_u32(tc_comm *comm, uint32_t *d_buf, size_t count);
tc_status comm_allgather(tc_comm *comm, void *d_buf, size_t bytes_per_rank);     // in place: rank r's part at r * bytes_per_rank
// Every rank calls it with the status of its own fallible set-up (allocations, index builds) BEFORE the first collective of a
// loop: one all-reduce of a flag; a rank that failed returns its own status, every other rank TC_GPU "a peer failed" -- nobody is
// left waiting in a collective its peer never enters.  One rank: returns `local`.
tc_status comm_agree(tc_comm *comm, tc_status local);

// icp.hip
tc_status icp_run_sharded(tc_context *ctx, tc_comm *comm, int shard_mode, bool p2plane, const float *d_src, size_t ns, const float *d_tgt,
                          size_t nt, const float *d_nrm, size_t nstride, const float init[7], size_t max_iters, float max_dist,
                          float conv_thr, tc_icp_result *res, DeviceIndex *tgt_prebuilt = nullptr);
// tgt_prebuilt: an index of the target built by the caller (a cloud handle; with its cell-sorted normals when p2plane), else
// the target is indexed into ctx->tgt_index; src_presorted: an index of the SOURCE the caller already has (a cloud handle that
// was indexed for its own normals): its cell-sorted records are walked as they are instead of sorting the source again
tc_status icp_run(tc_context *ctx, bool p2plane, const float *d_src, size_t ns, const float *d_tgt, size_t nt,
                  const float *d_nrm, size_t nstride, const float init[7], size_t max_iters,
                  float max_dist, float conv_thr, tc_icp_result *res, bool corr_on_device, int kiss = 0,
                  DeviceIndex *tgt_prebuilt = nullptr, const DeviceIndex *src_presorted = nullptr);
tc_status icp_run_gicp(tc_context *ctx, const float *d_src, size_t ns, const float *d_tgt, size_t nt, const float *d_cov_src,
                       const float *d_cov_tgt, const float init[7], size_t max_iters, float max_dist, float conv_thr,
                       tc_icp_result *res, bool corr_on_device);

}  // namespace tc
