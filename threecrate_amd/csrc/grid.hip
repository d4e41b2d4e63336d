// grid.hip -- device-built uniform-grid index (replaces KdTree::new, nearest_neighbor.rs:37-159).
//
// Pipeline (all on ctx->stream):
//   bbox_kernel      : min/max box (+ four sample boxes) folded into device accumulators with ordered-integer atomics
//   cell_hist_kernel : cell id per point + histogram
//   scan_*           : exclusive prefix sum of the histogram -> cell_start
//   place_kernel     : the record goes straight to cell_start[cell] + the arrival rank the histogram atomics returned
//                      (input read once, coalesced; atomic arrival order, not yet deterministic)
//   rerank_kernel    : streaming re-rank inside each cell by original index -> deterministic
//                      cell-sorted float4 {x, y, z, original-index bits}
//   (strict_order builds -- ranks that split one order between them -- replace the order of a cloud with an oversized cell by a
//    stable radix sort of (cell, original index) + rank_gather_kernel)
// HBM traffic per point: 12 B read (AoS xyz) + 4 B cell id + 16 B sorted record + 4 B slot
// (SURVEY 8d "index build" figure: 32 B/pt).
#include "tc_internal.h"
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace tc {

__device__ __forceinline__ void isometry_apply(const float q[4], const float t[3], float x, float y, float z,
                                               float &ox, float &oy, float &oz) {
    // nalgebra UnitQuaternion * Point3: t2 = 2 (qv x p); p' = t2*w + qv x t2 + p; then + translation
    float tx = (q[1] * z - q[2] * y) * 2.0f;
    float ty = (q[2] * x - q[0] * z) * 2.0f;
    float tz = (q[0] * y - q[1] * x) * 2.0f;
    float cx = q[1] * tz - q[2] * ty;
    float cy = q[2] * tx - q[0] * tz;
    float cz = q[0] * ty - q[1] * tx;
    ox = ((tx * q[3] + cx) + x) + t[0];
    oy = ((ty * q[3] + cy) + y) + t[1];
    oz = ((tz * q[3] + cz) + z) + t[2];
}

// per-block bounding box partials (6 floats per block); the host folds the <= 256 rows
// (min / max are order independent, so the result equals the reference's sequential fold).
constexpr int kBboxBlocks = 256;          // (round 6: 1024-thread blocks -- four times the reads in flight for the same 256 x 30 atomics -- are SLOWER, 22.8 vs 18.0 us)
// sbox (optional): four SAMPLE boxes per block.  Sample s = the points with index = s mod 4 whose multiplicative hash
// falls into one sixteenth of its range: ~n/64 points each, pseudo-random in the index (NOT every 64th point: organised
// scans are periodic in 64 -- beams, image columns -- and a sample must not be one beam).  A handful of far outliers
// shows up in the exact box but almost never in more than two of the four samples: the host compares them (cloud_bbox_impl).
// float <-> unsigned with the same order (atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); }
constexpr uint32_t kOrdPlusInf = 0xFF800000u, kOrdMinusInf = 0x007FFFFFu;       // f2ord(+inf), f2ord(-inf): the empty box
constexpr uint32_t kBboxAccStride = 32;        // words between two accumulators of the state block (ticket at word 0, accumulator k at (k + 1) * 32)
constexpr size_t kBboxStateBytes = 31 * kBboxAccStride * sizeof(uint32_t), kBboxBufBytes = kBboxStateBytes + 128 + 256;   // + the no-poll results

// The blocks fold their boxes into 30 accumulators in device memory with integer atomics (acc: [0..6) the exact box, [6..30) four
// sample boxes; `state`: ticket, then one accumulator per 128-byte line), drain them (s_waitcnt vmcnt(0)) and take a ticket; the LAST block reads the
// accumulators back (atomic loads: served where the atomics ran), stores the 30 floats to `out` -- the context's pinned HOST block,
// or device memory -- raises *done and resets the state for the next launch.  No per-block fence, no partials over PCIe (the first
// version of round 4 did both: 20.6 us instead of 10.9), no copy kernels, no stream synchronisation: the host polls *done.
// VEC (the cloud starts on a 16-byte boundary: every hipMalloc'd or torch-allocated tensor): a thread takes FOUR consecutive points
// per step as three 16-byte reads (perfectly coalesced; the 12-byte records read one float at a time cost three gathers per
// point), four steps in flight -- 1 M points are one round trip per thread instead of 16 dependent ones: 22 -> ~8 us.
template <bool VEC>
__global__ void __launch_bounds__(256) bbox_kernel(const float *__restrict__ xyz, uint32_t n, uint32_t *__restrict__ state, int robust,
                                                  float *__restrict__ out, uint32_t *__restrict__ done) {
    const bool sbox = robust != 0;
    __shared__ float sm[4][30];
    // v[0..6): the exact box (min xyz | max xyz); v[6 + 6 s ..): sample box s (points with index = s mod 4 whose hash is drawn)
    float v[30];
#pragma unroll
    for (int k = 0; k < 30; ++k) v[k] = (k % 6 < 3) ? INFINITY : -INFINITY;
    auto take = [&](float x, float y, float z, uint32_t i, int s) {
        // a point with a NaN or infinite coordinate takes no part in the box (it is indexed in the bucket behind the last cell)
        if (!(fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f)) return;
        v[0] = fminf(v[0], x); v[1] = fminf(v[1], y); v[2] = fminf(v[2], z);
        v[3] = fmaxf(v[3], x); v[4] = fmaxf(v[4], y); v[5] = fmaxf(v[5], z);
        if (sbox && ((i * 2654435761u) >> 28) == 0u) {
            float *b = v + 6 + 6 * s;
            b[0] = fminf(b[0], x); b[1] = fminf(b[1], y); b[2] = fminf(b[2], z);
            b[3] = fmaxf(b[3], x); b[4] = fmaxf(b[4], y); b[5] = fmaxf(b[5], z);
        }
    };
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if constexpr (VEC) {
        const float4 *__restrict__ q = reinterpret_cast<const float4 *>(xyz);
        const uint32_t nq = n >> 2;                 // whole groups of four points
        constexpr int U = 4;
        for (uint32_t c0 = gt; c0 < nq; c0 += U * stride) {
            float4 a[U], b[U], c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t cc = min(c0 + (uint32_t)u * stride, nq - 1u);       // (clamped: a repeated group changes no box)
                a[u] = q[3 * (size_t)cc]; b[u] = q[3 * (size_t)cc + 1]; c[u] = q[3 * (size_t)cc + 2];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t i = 4u * min(c0 + (uint32_t)u * stride, nq - 1u);
                take(a[u].x, a[u].y, a[u].z, i, 0);
                take(a[u].w, b[u].x, b[u].y, i + 1u, 1);
                take(b[u].z, b[u].w, c[u].x, i + 2u, 2);
                take(c[u].y, c[u].z, c[u].w, i + 3u, 3);
            }
        }
        if (gt < (n & 3u)) {                         // the last n mod 4 points
            const uint32_t i = (nq << 2) + gt;
            const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
            if ((i & 3u) == 0u) take(x, y, z, i, 0); else if ((i & 3u) == 1u) take(x, y, z, i, 1); else take(x, y, z, i, 2);
        }
    } else {
        // each thread reads whole points; consecutive lanes read consecutive 12-B records; i = lane mod 4 for every point of a
        // thread (the stride is a multiple of 64)
        for (uint32_t i = gt; i < n; i += stride) {
            const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
            const uint32_t sidx = threadIdx.x & 3u;
            if (sidx == 0u) take(x, y, z, i, 0); else if (sidx == 1u) take(x, y, z, i, 1); else if (sidx == 2u) take(x, y, z, i, 2); else take(x, y, z, i, 3);
        }
    }
    // fold over the wave (min / max: order independent), the four waves through LDS, then the block's 30 values into the accumulators
#pragma unroll
    for (int k = 0; k < 30; ++k) {
        if (k < 6 || sbox) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float w = __shfl_xor(v[k], o);
                v[k] = (k % 6 < 3) ? fminf(v[k], w) : fmaxf(v[k], w);
            }
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 30; ++k) sm[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    // (one accumulator per 128-byte line: 256 blocks x 30 atomics on ONE line serialise at the memory side -- the kernel took 21 us
    // whatever the reads cost; on 30 lines they run side by side)
    uint32_t *acc = state + kBboxAccStride;
    if (threadIdx.x < (sbox ? 30 : 6)) {
        const bool is_min = threadIdx.x % 6 < 3;
        float r = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) r = is_min ? fminf(r, sm[w][threadIdx.x]) : fmaxf(r, sm[w][threadIdx.x]);
        if (is_min) atomicMin(&acc[threadIdx.x * kBboxAccStride], f2ord(r)); else atomicMax(&acc[threadIdx.x * kBboxAccStride], f2ord(r));
    }
    __shared__ uint32_t s_last;
    if (threadIdx.x < 64) {                       // the lanes that issued atomics are in wave 0: its atomics are done before its ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 30) {
        const uint32_t u = __hip_atomic_load(&acc[threadIdx.x * kBboxAccStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out[threadIdx.x], ord2f(u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&acc[threadIdx.x * kBboxAccStride], (threadIdx.x % 6 < 3) ? kOrdPlusInf : kOrdMinusInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) __hip_atomic_store(state, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done) {
        if (threadIdx.x < 30) __threadfence_system();            // (one block, once: the 30 stores are out before the flag)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void bbox_state_init_kernel(uint32_t *__restrict__ state) {
    const uint32_t t = threadIdx.x;
    if (t == 0) state[0] = 0u;
    if (t < 30) state[(t + 1) * kBboxAccStride] = (t % 6 < 3) ? kOrdPlusInf : kOrdMinusInf;
}


// cell id per point (+ histogram).  With `st` != null the point is first moved by the
// isometry in *st (ICP source ordering by target cell); the stored record keeps the raw point.
__global__ void __launch_bounds__(256) cell_hist_kernel(const float *__restrict__ xyz, uint32_t n, GridGeom g,
                                                       const IcpState *__restrict__ st, TileGeom tg, int tile_major, uint32_t nkeys,
                                                       uint32_t *__restrict__ cell_of, uint32_t *__restrict__ hist,
                                                       uint32_t *__restrict__ arrival, uint32_t *__restrict__ pts_pad,
                                                       uint32_t *__restrict__ cs_front, uint32_t *__restrict__ cs_tail, uint32_t rank_max) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // the paddings the ICP search over-reads (three tiny memsets = three launches otherwise): huge coordinates behind the
    // sorted records, zeros around the prefix sums
    if (blockIdx.x == 0 && threadIdx.x < 4 * kPtsPad + kCellStartFront + kCellStartPad) {
        const uint32_t t = threadIdx.x;
        if (t < 4 * kPtsPad) pts_pad[t] = 0x7F7F7F7Fu;
        else if (t < 4 * kPtsPad + kCellStartFront) cs_front[t - 4 * kPtsPad] = 0u;
        else cs_tail[t - 4 * kPtsPad - kCellStartFront] = 0u;
    }
    if (i >= n) return;
    float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    if (st) {
        float q[4] = {st->q[0], st->q[1], st->q[2], st->q[3]}, t[3] = {st->t[0], st->t[1], st->t[2]};
        float ox, oy, oz;
        isometry_apply(q, t, x, y, z, ox, oy, oz);
        x = ox; y = oy; z = oz;
    }
    int ix = cell_coord(x, g.minx, g.inv_h, g.gx);
    int iy = cell_coord(y, g.miny, g.inv_h, g.gy);
    int iz = cell_coord(z, g.minz, g.inv_h, g.gz);
    uint32_t c = tile_major ? tile_major_id(tg, ix, iy, iz) : ((uint32_t)iz * g.gy + iy) * g.gx + ix;
    // Non-finite points (NaN pixels of an organised depth image, inf ranges) live in ONE extra bucket behind the last cell:
    // no neighbour search ever visits it (a scan ends at cell_start[ncell]), so they are inert as candidates, and their
    // records hold huge finite coordinates (rank_gather_kernel) -- harmless when a 4-wide step reads past its span.
    if (!(fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f)) c = nkeys;
    cell_of[i] = c;
    // the returned count is this point's arrival rank inside its cell: the scatter pass then needs
    // no second round of atomics
    const uint32_t a = atomicAdd(&hist[c], 1u);
    arrival[i] = a;
    // a cell too populous for the quadratic re-rank (rank_gather_kernel): the word behind the histogram says so
    // (build_index(strict_order): callers whose RANKS must agree on the order re-sort then)
    if (a == rank_max) hist[nkeys + 1] = 1u;
}

__global__ void __launch_bounds__(256) iota_kernel(uint32_t *__restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

constexpr int kScanItems = 8;
constexpr int kScanBlock = 256;
constexpr int kScanTile = kScanItems * kScanBlock;

// per-block sums, and per-block counts of non-zero entries behind them (blocksum[nblk + b]): the number of
// OCCUPIED cells comes for free with the prefix sum (build_index adapts the cell edge with it)
__global__ void __launch_bounds__(kScanBlock) scan_reduce_kernel(const uint32_t *__restrict__ in, uint32_t n,
                                                                 uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t wsum[kScanBlock / 64], wnz[kScanBlock / 64];
    uint32_t base = blockIdx.x * kScanTile;
    uint32_t s = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t i = base + k * kScanBlock + threadIdx.x;
        if (i < n) { const uint32_t v = in[i]; s += v; nz += v != 0; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); nz += __shfl_xor(nz, o); }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wnz[threadIdx.x >> 6] = nz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        blocksum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        blocksum[gridDim.x + blockIdx.x] = wnz[0] + wnz[1] + wnz[2] + wnz[3];
    }
}

// (the exclusive scan of the block sums used to be a launch of its own, ~4.6 us of a 16 us scan: every apply block now adds up
// the sums in front of it itself -- a few hundred values --, block 0 also the non-zero counts -> blocksum[2 * nb])
// (that re-summing is quadratic in the number of scan blocks: fine for the few hundred of a uniform 1 M-point grid, 5.8 us;
// an adapted grid -- up to 32 cells per point -- of 16 M points has 250 k of them = 3e10 loads.  Beyond kScanFusedMax blocks a
// one-block kernel turns the block sums into their exclusive prefix first and the apply blocks read theirs: `prefixed`.)
constexpr uint32_t kScanFusedMax = 2048;
__global__ void __launch_bounds__(1024) scan_top_kernel(uint32_t *__restrict__ blocksum, uint32_t nb) {
    __shared__ uint32_t wtot[16];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? blocksum[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if ((threadIdx.x & 63) >= (unsigned)o) inc += t;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < (int)(threadIdx.x >> 6)) woff += wtot[w]; tot += wtot[w]; }
        if (i < nb) blocksum[i] = carry + woff + inc - v;
        carry += tot;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kScanBlock) scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t n,
                                                                uint32_t *__restrict__ blocksum,
                                                                uint32_t *__restrict__ out /* n+1 */, uint32_t *__restrict__ occ_out, int prefixed,
                                                                unsigned long long *__restrict__ occ_host) {
    __shared__ uint32_t wtot[kScanBlock / 64];
    __shared__ uint32_t wpre[kScanBlock / 64], wnz[kScanBlock / 64];
    uint32_t pre = 0, nzt = 0;
    if (prefixed) { if (threadIdx.x == 0) pre = blocksum[blockIdx.x]; }
    else for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kScanBlock) pre += blocksum[i];
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < gridDim.x; i += kScanBlock) nzt += blocksum[gridDim.x + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); nzt += __shfl_xor(nzt, o); }
    if ((threadIdx.x & 63) == 0) { wpre[threadIdx.x >> 6] = pre; wnz[threadIdx.x >> 6] = nzt; }
    // thread owns kScanItems CONSECUTIVE items so the in-thread prefix is sequential
    uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t i = base + k;
        v[k] = (i < n) ? in[i] : 0;
        s += v[k];
    }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o);
        if ((threadIdx.x & 63) >= (unsigned)o) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wtot[w];
    uint32_t block_pre = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) block_pre += wpre[w];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) t += wnz[w];
        blocksum[2 * gridDim.x] = t;
        if (occ_out) *occ_out = t;          // (build_index: kept in front of the prefix sums -- the normals kernel picks its path by it)
        // (build_index's edge adaptation: count + flag in ONE 8-byte word of the pinned host block, which the host polls while the
        // rest of the build runs -- no copy kernel, no stream synchronisation behind the build)
        if (occ_host) __hip_atomic_store(occ_host, (1ull << 32) | (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    uint32_t run = block_pre + woff + inc - s;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
        if (i == n - 1) out[n] = run;
    }
}

// (strict-order path only) records gathered through `slot`, the original indices in final order (slot_is_ordered), or --
// historically -- in arrival order with a stable re-rank inside the cell
__global__ void __launch_bounds__(256) rank_gather_kernel(const float *__restrict__ xyz, uint32_t n, uint32_t nkeys,
                                                         const uint32_t *__restrict__ cell_of,
                                                         const uint32_t *__restrict__ cell_start,
                                                         const uint32_t *__restrict__ slot,
                                                         float4 *__restrict__ pts, int slot_is_ordered, uint32_t rank_max) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t i = slot[p];
    uint32_t c = cell_of[i];
    uint32_t s = cell_start[c], e = cell_start[c + 1];
    // rank = the number of points of the cell with a smaller original index: quadratic in the cell's population, which is ~1.5 on
    // the clouds this index is sized for.  A cluster far below the cell edge or a block of exact duplicates pays for it -- 120 k
    // points in ONE cell rank in ~1.5 ms, 2^20 in ~0.1 s -- which is what ONE neighbour-search pass over such a cell costs as
    // well (every query scans the whole cell).  Beyond 2^20 points in a cell the atomic arrival order is kept -- the only place
    // where two runs may differ.  (The cut-off was 65 536 until round 4: a 120 k-point cluster of sigma =
    // 0.002 in a unit cloud made one normal in ~10 runs differ, by an equidistant neighbour pair -- tools/dev/index_stress.py.)
    uint32_t rank;
    if (slot_is_ordered) {            // `slot` comes from a stable sort by cell: already in ascending original index
        rank = p - s;
    } else if (e - s <= rank_max) {
        rank = 0;
        for (uint32_t j = s; j < e; ++j) rank += (slot[j] < i) ? 1u : 0u;
    } else {
        rank = p - s;
    }
    float4 r;
    r.x = xyz[3 * (size_t)i]; r.y = xyz[3 * (size_t)i + 1]; r.z = xyz[3 * (size_t)i + 2];
    if (c == nkeys) r.x = r.y = r.z = __uint_as_float(0x7F7F7F7Fu);     // the non-finite bucket: d2 to anything = +inf, never selected
    r.w = __uint_as_float(i);
    pts[s + rank] = r;
}

// Direct placement (round 3): the point's record goes straight to  cell_start[cell] + arrival rank  -- the input is read once,
// coalesced, instead of being gathered 12 bytes at a time through the permutation (7x over-fetch) -- and a second, streaming
// pass re-ranks the records INSIDE each cell by original index (the deterministic order; the arrival order is the atomics').
__global__ void __launch_bounds__(256) place_kernel(const float *__restrict__ xyz, uint32_t n, uint32_t nkeys,
                                                   const uint32_t *__restrict__ cell_of, const uint32_t *__restrict__ cell_start,
                                                   const uint32_t *__restrict__ arrival, float4 *__restrict__ tmp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = cell_of[i];
    const uint32_t pos = cell_start[c] + arrival[i];
    float4 r;
    r.x = xyz[3 * (size_t)i]; r.y = xyz[3 * (size_t)i + 1]; r.z = xyz[3 * (size_t)i + 2];
    if (c == nkeys) r.x = r.y = r.z = __uint_as_float(0x7F7F7F7Fu);     // the non-finite bucket: d2 to anything = +inf, never selected
    r.w = __uint_as_float(i);
    tmp[pos] = r;
}

// (the record's cell is computed again from its coordinates -- the same expressions on the same bits as cell_hist_kernel -- rather
// than carried in a second scattered array: scattered 4-byte writes cost as much as the 16-byte ones)
__global__ void __launch_bounds__(256) rerank_kernel(uint32_t n, GridGeom g, const IcpState *__restrict__ st, TileGeom tg, int tile_major,
                                                    uint32_t nkeys, const uint32_t *__restrict__ cell_start, const float4 *__restrict__ tmp,
                                                    float4 *__restrict__ pts, uint32_t rank_max) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 r = tmp[p];
    uint32_t c;
    {
        float x = r.x, y = r.y, z = r.z;
        if (st) {
            float q[4] = {st->q[0], st->q[1], st->q[2], st->q[3]}, t[3] = {st->t[0], st->t[1], st->t[2]};
            float ox, oy, oz;
            isometry_apply(q, t, x, y, z, ox, oy, oz);
            x = ox; y = oy; z = oz;
        }
        const int ix = cell_coord(x, g.minx, g.inv_h, g.gx), iy = cell_coord(y, g.miny, g.inv_h, g.gy), iz = cell_coord(z, g.minz, g.inv_h, g.gz);
        c = tile_major ? tile_major_id(tg, ix, iy, iz) : ((uint32_t)iz * g.gy + iy) * g.gx + ix;
        // (the non-finite bucket's records hold the placeholder 0x7F7F7F7F = 3.39e38; a transformed finite point that overflowed
        // was put there by cell_hist_kernel through the same test)
        if (!(fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f)) c = nkeys;
    }
    const uint32_t s = cell_start[c], e = cell_start[c + 1];
    const uint32_t i = __float_as_uint(r.w);
    // rank = the number of records of the cell with a smaller original index (see rank_gather_kernel for the population cut-off)
    uint32_t rank;
    if (e - s <= rank_max) {
        rank = 0;
#pragma unroll 4
        for (uint32_t j = s; j < e; ++j) rank += (__float_as_uint(tmp[j].w) < i) ? 1u : 0u;
    } else {
        rank = p - s;
    }
    pts[s + rank] = r;
}

__global__ void __launch_bounds__(256) gather_normals_kernel(const float4 *__restrict__ pts, uint32_t n,
                                                            const float *__restrict__ nrm, size_t stride,
                                                            float4 *__restrict__ out) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t i = __float_as_uint(pts[p].w);
    const float *s = nrm + (size_t)i * stride;
    out[p] = make_float4(s[0], s[1], s[2], 0.0f);
}

GridView view_of(const DeviceIndex &ix) {
    GridView v;
    v.g = ix.geom;
    v.pts = (const float4 *)ix.pts.p;
    v.cell_start = (const uint32_t *)ix.cell_start.p + kCellStartFront;
    v.pts12 = ix.pts12_valid ? (const float *)ix.pts12.p : nullptr;
    return v;
}

// Cell edge from the bounding box: h = f * (measure / n)^(1/d) over the non-degenerate axes.
// grid dimensions for cell edge h over the box already stored in g; h grows until the cell budget holds
static void set_cell_edge(GridGeom &g, double h, size_t n, double cells_per_point) {
    const double e[3] = {(double)g.maxx - g.minx, (double)g.maxy - g.miny, (double)g.maxz - g.minz};
    const double max_cells = std::max<double>(cells_per_point * (double)n, 4096.0);   // dense grid: cells cost memory and scan time
    for (int guard = 0; guard < 200; ++guard) {
        double gx = std::floor(e[0] / h) + 1.0, gy = std::floor(e[1] / h) + 1.0, gz = std::floor(e[2] / h) + 1.0;
        if (gx * gy * gz <= max_cells && gx * gy * gz < 2.0e9) {
            g.gx = (int)gx; g.gy = (int)gy; g.gz = (int)gz;
            break;
        }
        h *= 1.2;
    }
    g.h = (float)h;
    g.inv_h = 1.0f / g.h;
    g.ncell = (uint32_t)g.gx * (uint32_t)g.gy * (uint32_t)g.gz;
    g.n = (uint32_t)n;
}

static void derive_geom(GridGeom &g, const float mn[3], const float mx[3], size_t n, float f, float min_h) {
    g.minx = mn[0]; g.miny = mn[1]; g.minz = mn[2];
    g.maxx = mx[0]; g.maxy = mx[1]; g.maxz = mx[2];
    g.cx = 0.5f * (mn[0] + mx[0]); g.cy = 0.5f * (mn[1] + mx[1]); g.cz = 0.5f * (mn[2] + mx[2]);
    double e[3] = {(double)mx[0] - mn[0], (double)mx[1] - mn[1], (double)mx[2] - mn[2]};
    double emax = std::max(e[0], std::max(e[1], e[2]));
    double h = 1.0;
    if (!(emax > 0.0) || !std::isfinite(emax)) {
        h = 1.0;
    } else {
        double measure = 1.0;
        int dims = 0;
        for (int a = 0; a < 3; ++a)
            if (e[a] > 1e-6 * emax) { measure *= e[a]; ++dims; }
        h = f * std::pow(measure / (double)std::max<size_t>(n, 1), 1.0 / (double)dims);
        if (!(h > 0.0) || !std::isfinite(h)) h = emax;
        h = std::max(h, emax * 1e-4);   // at most 10^4 cells per axis
    }
    if (min_h > 0.0f && std::isfinite(min_h)) h = std::max(h, (double)min_h);
    set_cell_edge(g, h, n, 8.0);
}

TileGeom make_tiles(const GridGeom &g, int tx, int ty, int tz) {
    TileGeom t;
    t.tx = std::max(tx, 1); t.ty = std::max(ty, 1); t.tz = std::max(tz, 1);
    t.ntx = (g.gx + t.tx - 1) / t.tx; t.nty = (g.gy + t.ty - 1) / t.ty; t.ntz = (g.gz + t.tz - 1) / t.tz;
    t.cpt = (uint32_t)(t.tx * t.ty * t.tz);
    t.ntiles = (uint32_t)t.ntx * (uint32_t)t.nty * (uint32_t)t.ntz;
    return t;
}

tc_status exclusive_scan_u32(tc_context *ctx, const uint32_t *d_in, uint32_t n, uint32_t *d_out, DevBuf &blocksum, uint32_t *occ_out,
                             unsigned long long *occ_host) {
    hipStream_t st = ctx->stream;
    const uint32_t nscan = (n + kScanTile - 1) / kScanTile;
    if (tc_status s = ensure(ctx, blocksum, ((size_t)2 * nscan + 1) * sizeof(uint32_t))) return s;   // sums | non-zero counts | their total
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nscan), dim3(kScanBlock), 0, st, d_in, n, (uint32_t *)blocksum.p);
    // (TC_SCAN_FUSED_MAX, read per call: the tests force the two-level path on small grids)
    const char *fm = getenv("TC_SCAN_FUSED_MAX");
    const bool prefixed = nscan > (fm ? (uint32_t)atoi(fm) : kScanFusedMax);
    if (prefixed) hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, st, (uint32_t *)blocksum.p, nscan);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nscan), dim3(kScanBlock), 0, st, d_in, n, (uint32_t *)blocksum.p, d_out, occ_out, prefixed ? 1 : 0, occ_host);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

// exact box in mn / mx; with `rmn` also the box the grid should span: per axis the exact range unless two of the
// four sample boxes agree that it is more than 1.3 x wider than the cloud proper (far outliers: a flying pixel, a
// stray return), in which case the sampled range + 5 % -- points outside are indexed in the boundary cells.
static tc_status cloud_bbox_impl(tc_context *ctx, const float *d_xyz, size_t n, float mn[3], float mx[3], float *rmn, float *rmx,
                                 bool *clamped) {
    hipStream_t st = ctx->stream;
    const int nb = (int)((n + 255) / 256);
    const int bb = std::min(nb, kBboxBlocks);
    const bool robust = rmn != nullptr && n >= 4096;
    // The blocks fold into device accumulators, the last one stores the 30 floats (exact box + four sample boxes) into the pinned
    // host block and raises a word there; the host polls it: no copy kernels, no stream synchronisation (TC_NO_PINNED_POLL=1: the
    // result stays on the device and comes back through a copy + synchronisation, as in rounds 1-3).
    const bool poll = pinned_poll_enabled();
    const bool fresh = ctx->bbox.p == nullptr;
    if (tc_status s = ensure(ctx, ctx->bbox, kBboxBufBytes)) return s;
    uint32_t *d_state = (uint32_t *)ctx->bbox.p;                       // ticket | 30 accumulators, a 128-byte line each | (no-poll: 30 results behind them)
    if (fresh) hipLaunchKernelGGL(bbox_state_init_kernel, dim3(1), dim3(64), 0, st, d_state);
    float *hb = (float *)((char *)ctx->pinned + 2048);                  // [0..6) exact box, [6..30) sample boxes
    volatile uint32_t *h_done = (volatile uint32_t *)((char *)ctx->pinned + 2048 + 8192 + 192);
    float *d_out = (float *)((char *)ctx->bbox.p + kBboxStateBytes + 128);
    uint32_t *d_done = nullptr;
    if (poll) {
        d_out = (float *)pinned_dev_ptr(ctx, hb);
        d_done = (uint32_t *)pinned_dev_ptr(ctx, (const void *)h_done);
        if (!d_out || !d_done) return fail(ctx, TC_GPU, "bbox: the pinned block has no device address");
        *h_done = 0u;
    }
    {
        ProfScope ps(ctx, "bbox");
        if (((uintptr_t)d_xyz & 15u) == 0u)
            hipLaunchKernelGGL(bbox_kernel<true>, dim3(std::min((int)((n / 4 + 255) / 256) + 1, kBboxBlocks)), dim3(256), 0, st, d_xyz, (uint32_t)n, d_state, robust ? 1 : 0, d_out, d_done);
        else
            hipLaunchKernelGGL(bbox_kernel<false>, dim3(bb), dim3(256), 0, st, d_xyz, (uint32_t)n, d_state, robust ? 1 : 0, d_out, d_done);
    }
    TC_HIP_TRY(ctx, hipGetLastError());
    if (poll) {
        if (tc_status s = wait_pinned_word(ctx, h_done, "bounding box")) { hipLaunchKernelGGL(bbox_state_init_kernel, dim3(1), dim3(64), 0, st, d_state); return s; }
    } else {
        TC_HIP_TRY(ctx, hipMemcpyAsync(hb, d_out, 30 * sizeof(float), hipMemcpyDeviceToHost, st));
        TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    const float *hs = hb + 6;
    for (int c = 0; c < 3; ++c) { mn[c] = INFINITY; mx[c] = -INFINITY; }
    for (int c = 0; c < 3; ++c) { mn[c] = std::fmin(mn[c], hb[c]); mx[c] = std::fmax(mx[c], hb[3 + c]); }
    if (rmn) {
        for (int c = 0; c < 3; ++c) { rmn[c] = mn[c]; rmx[c] = mx[c]; }
        *clamped = false;
    }
    if (!robust) return TC_OK;
    float acc[24];                                   // [sample][min xyz | max xyz], folded over the blocks (plain compares: no NaNs in here)
    for (int i = 0; i < 24; ++i) acc[i] = (i % 6 < 3) ? INFINITY : -INFINITY;
    for (int i = 0; i < 24; ++i) acc[i] = (i % 6 < 3) ? (hs[i] < acc[i] ? hs[i] : acc[i]) : (hs[i] > acc[i] ? hs[i] : acc[i]);
    for (int c = 0; c < 3; ++c) {
        float lo[4], hi[4];
        for (int k = 0; k < 4; ++k) { lo[k] = acc[6 * k + c]; hi[k] = acc[6 * k + 3 + c]; }
        std::sort(lo, lo + 4);                       // ascending: contaminated samples first
        std::sort(hi, hi + 4, [](float a, float b) { return a > b; });
        const float slo = lo[2], shi = hi[2];        // tolerate two contaminated samples per side
        if (!(slo <= shi) || !std::isfinite(slo) || !std::isfinite(shi)) continue;
        const float ext = shi - slo, full = mx[c] - mn[c];
        if (!(full > 1.3f * ext) || !(full > 0.0f)) continue;
        rmn[c] = std::fmax(mn[c], slo - 0.05f * ext);
        rmx[c] = std::fmin(mx[c], shi + 0.05f * ext);
        *clamped = true;
    }
    return TC_OK;
}

tc_status cloud_bbox(tc_context *ctx, const float *d_xyz, size_t n, float mn[3], float mx[3]) {
    return cloud_bbox_impl(ctx, d_xyz, n, mn, mx, nullptr, nullptr, nullptr);
}


// ---- binned placement (round 4) ------------------------------------------------------------------------------------------------
// The counting sort above pays 1 M scattered returning atomics for 1 M points (cell_hist_kernel: 42 us, the memory side's rate
// for 64 lanes in 64 different rows), a scattered 16-byte placement (23 us) and a re-rank pass (10 us).  On a grid that is dense in
// the cloud's sense -- at most ~8 cells per point -- the same layout (cell-sorted, ascending original index inside a cell: bit for
// bit the layout of the passes above) comes out of two LDS stages without a global atomic per point:
//   bin_count / bin_offsets : 256 blocks walk contiguous shares of the input and count, in LDS, how many of their points fall
//        into each BIN (a run of consecutive keys holding ~2048 points); per-(block, bin) offsets + the bins' totals;
//   bin_scatter  : every block scans the totals (the bins' starts), then the same walk again: each record goes to its block's run
//        inside its bin (LDS cursor: one returning LDS atomic);
//   bin_place    : one block per bin -- its records' keys counted in LDS, scanned (-> cell_start of the bin's keys), every record
//        ranked inside its cell by original index (LDS), the bin written out in final order with coalesced stores.
// A bin that does not fit a block's LDS (kBinCap records: a cloud far denser in one place than on average) sends the whole build
// back to the atomic passes: the host reads the largest bin's population (one polled word, written by block 0 of bin_scatter).
constexpr int kBinBlocks = 256;               // blocks of the count / scatter passes
constexpr int kBinWalkThreads = 1024;         // their threads (the walks are latency bound: 65 k threads left 15 dependent rounds each)
#ifndef TC_BIN_TARGET
#define TC_BIN_TARGET 2048
#endif
#ifndef TC_BIN_PLACE_THREADS
#define TC_BIN_PLACE_THREADS 512
#endif
constexpr uint32_t kBinTarget = TC_BIN_TARGET;          // points per bin aimed at
constexpr uint32_t kBinCap = 2 * kBinTarget;            // records one bin_place block holds in LDS
constexpr uint32_t kBinKeysMax = 2 * kBinTarget;        // keys (cells) per bin: one LDS counter each
// bin_place_kernel holds cnt + rkey + ridx + slot (+ a few words) in static LDS: 64 KB and a bit -- fine on gfx950 (160 KB per CU),
// not on a 64 KB-LDS target: this library is gfx950 only (Makefile: ARCH), say so here rather than in a linker error
static_assert((kBinKeysMax + 3 * kBinCap) * sizeof(uint32_t) + 256 <= 160 * 1024, "bin_place_kernel's static LDS exceeds a gfx950 CU's 160 KB");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libthreecrate_hip is written for gfx950 (MI355X): bin_place_kernel alone needs more than 64 KB of LDS per workgroup"
#endif
constexpr uint32_t kBinMaxBins = 8192;                  // LDS counters of the count / scatter passes
constexpr int kBinPlaceThreads = TC_BIN_PLACE_THREADS;

__device__ __forceinline__ uint32_t point_key(float x, float y, float z, const GridGeom &g, const IcpState *__restrict__ st, const TileGeom &tg,
                                              int tile_major, uint32_t nkeys) {
    // (the same expressions on the same bits as cell_hist_kernel / rerank_kernel)
    if (st) {
        float q[4] = {st->q[0], st->q[1], st->q[2], st->q[3]}, t[3] = {st->t[0], st->t[1], st->t[2]};
        float ox, oy, oz;
        isometry_apply(q, t, x, y, z, ox, oy, oz);
        x = ox; y = oy; z = oz;
    }
    const int ix = cell_coord(x, g.minx, g.inv_h, g.gx), iy = cell_coord(y, g.miny, g.inv_h, g.gy), iz = cell_coord(z, g.minz, g.inv_h, g.gz);
    uint32_t c = tile_major ? tile_major_id(tg, ix, iy, iz) : ((uint32_t)iz * g.gy + iy) * g.gx + ix;
    if (!(fabsf(x) <= 3.0e38f && fabsf(y) <= 3.0e38f && fabsf(z) <= 3.0e38f)) c = nkeys;
    return c;
}

__global__ void __launch_bounds__(kBinWalkThreads) bin_count_kernel(const float *__restrict__ xyz, uint32_t n, GridGeom g, const IcpState *__restrict__ st,
                                                       TileGeom tg, int tile_major, uint32_t nkeys, uint32_t kpb, uint32_t nbins,
                                                       uint32_t *__restrict__ cnt /* [nbins][kBinBlocks] */, uint32_t *__restrict__ pts_pad,
                                                       uint32_t *__restrict__ cs_front, uint32_t *__restrict__ cs_tail) {
    __shared__ uint32_t c[kBinMaxBins];
    if (blockIdx.x == 0 && threadIdx.x < 4 * kPtsPad + kCellStartFront + kCellStartPad) {     // the paddings (see cell_hist_kernel)
        const uint32_t t = threadIdx.x;
        if (t < 4 * kPtsPad) pts_pad[t] = 0x7F7F7F7Fu;
        else if (t < 4 * kPtsPad + kCellStartFront) cs_front[t - 4 * kPtsPad] = 0u;
        else cs_tail[t - 4 * kPtsPad - kCellStartFront] = 0u;
    }
    for (uint32_t b = threadIdx.x; b < nbins; b += kBinWalkThreads) c[b] = 0u;
    __syncthreads();
    const uint32_t per = (n + kBinBlocks - 1) / kBinBlocks;
    const uint32_t i0 = blockIdx.x * per, i1 = min(i0 + per, n);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += kBinWalkThreads) {
        const uint32_t k = point_key(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], g, st, tg, tile_major, nkeys);
        atomicAdd(&c[min(k / kpb, nbins - 1u)], 1u);          // (k <= nkeys < nbins * kpb; the clamp only keeps a key that broke that rule inside LDS)
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += kBinWalkThreads) cnt[(size_t)b * kBinBlocks + blockIdx.x] = c[b];        // [bin][block]
}

// per bin: exclusive prefix over the blocks (in place) + the bin's total.  One wave per bin: a lane takes four consecutive blocks
// (one 16-byte read), in-lane prefix + wave scan.  cnt is [bin][kBinBlocks].
static_assert(kBinBlocks == 256, "one wave x four entries per lane");
__global__ void __launch_bounds__(256) bin_offsets_kernel(uint32_t *__restrict__ cnt, uint32_t nbins, uint32_t *__restrict__ tot) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (b >= nbins) return;
    uint4 *row = reinterpret_cast<uint4 *>(cnt + (size_t)b * kBinBlocks);
    const uint4 v = row[lane];
    const uint32_t s = v.x + v.y + v.z + v.w;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= (unsigned)o) inc += t;
    }
    const uint32_t ex = inc - s;
    row[lane] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
    if (lane == 63u) tot[b] = inc;
}

// Every block first scans the bin totals itself (nbins <= 8192 words: cheaper than the 4.7 us floor of a one-block launch in
// between, which this was until the end of round 4): bin starts = exclusive scan of the totals.  Block 0 also writes them out for
// bin_place (nbins + 1 entries), zeroes its {occupied cells, ticket} pair and publishes the largest bin's population twice: in
// device memory for bin_place, which is enqueued right behind this kernel and leaves at once when a bin does not fit a block's
// LDS, and in the pinned host word (flag << 32 | population), from which the host learns whether the atomic passes must follow.
__global__ void __launch_bounds__(kBinWalkThreads) bin_scatter_kernel(const float *__restrict__ xyz, uint32_t n, GridGeom g, const IcpState *__restrict__ st,
                                                         TileGeom tg, int tile_major, uint32_t nkeys, uint32_t kpb, uint32_t nbins,
                                                         const uint32_t *__restrict__ off /* [nbins][kBinBlocks] */,
                                                         const uint32_t *__restrict__ tot, uint32_t *__restrict__ binstart,
                                                         uint32_t *__restrict__ occ_ticket, unsigned long long *__restrict__ host_word,
                                                         float4 *__restrict__ tmp) {
    static_assert(kBinWalkThreads == 1024, "the scan below is written for 16 waves");
    __shared__ uint32_t cur[kBinMaxBins];
    __shared__ uint32_t wtot[16], wmax[16];
    uint32_t carry = 0, mx = 0;
    for (uint32_t base = 0; base < nbins; base += kBinWalkThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nbins ? tot[i] : 0u;
        const uint32_t o_mine = i < nbins ? off[(size_t)i * kBinBlocks + blockIdx.x] : 0u;
        mx = max(mx, v);
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if ((threadIdx.x & 63) >= (unsigned)o) inc += t;
        }
        if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < (int)(threadIdx.x >> 6)) woff += wtot[w]; total += wtot[w]; }
        const uint32_t start = carry + woff + inc - v;
        if (i < nbins) {
            cur[i] = start + o_mine;
            if (blockIdx.x == 0) binstart[i] = start;
        }
        carry += total;
        __syncthreads();
    }
    if (blockIdx.x == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            binstart[nbins] = carry;
            uint32_t m = 0;
            for (int w = 0; w < 16; ++w) m = max(m, wmax[w]);
            occ_ticket[0] = 0u; occ_ticket[1] = 0u;
            occ_ticket[2] = m;                       // bin_place reads it: launched before the host has looked at the word below
            __hip_atomic_store(host_word, (1ull << 32) | (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const uint32_t per = (n + kBinBlocks - 1) / kBinBlocks;
    const uint32_t i0 = blockIdx.x * per, i1 = min(i0 + per, n);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += kBinWalkThreads) {
        float4 r;
        r.x = xyz[3 * (size_t)i]; r.y = xyz[3 * (size_t)i + 1]; r.z = xyz[3 * (size_t)i + 2];
        const uint32_t k = point_key(r.x, r.y, r.z, g, st, tg, tile_major, nkeys);
        if (k == nkeys) r.x = r.y = r.z = __uint_as_float(0x7F7F7F7Fu);     // the non-finite bucket (see place_kernel)
        r.w = __uint_as_float(i);
        tmp[atomicAdd(&cur[min(k / kpb, nbins - 1u)], 1u)] = r;
    }
}

__global__ void __launch_bounds__(kBinPlaceThreads) bin_place_kernel(const float4 *__restrict__ tmp, const uint32_t *__restrict__ binstart,
                                                                     uint32_t nbins, uint32_t kpb, uint32_t nkeys, GridGeom g,
                                                                     const IcpState *__restrict__ st, TileGeom tg, int tile_major,
                                                                     uint32_t *__restrict__ cs, float4 *__restrict__ pts,
                                                                     uint32_t *__restrict__ occ_ticket, uint32_t *__restrict__ occ_out,
                                                                     unsigned long long *__restrict__ occ_host) {
    __shared__ uint32_t cnt[kBinKeysMax];            // per key of the bin: count, then exclusive start
    __shared__ uint32_t rkey[kBinCap];               // record i: local key | arrival rank << 16, later its final slot
    __shared__ uint32_t ridx[kBinCap];               // record i: original index
    __shared__ uint32_t slot[kBinCap];               // by (start + arrival): original index; later by final slot: record
    __shared__ uint32_t wsum[kBinPlaceThreads / 64];
    __shared__ uint32_t s_occ;
    const uint32_t bin = blockIdx.x, tid = threadIdx.x;
    // The launch is enqueued right behind the scatter pass, before the host knows the bins' populations (no bubble on the stream
    // for its decision): when a bin does not fit a block's LDS every block leaves here, nothing written, and the host -- which
    // reads the same number from its pinned word -- enqueues the atomic counting sort.
    if (occ_ticket[2] > kBinCap) return;
    const uint32_t bs = binstart[bin], P = binstart[bin + 1] - bs;
    const uint32_t k0 = bin * kpb, K = min(kpb, nkeys + 1u - k0);          // this bin's keys: k0 .. k0 + K
    for (uint32_t k = tid; k < K; k += kBinPlaceThreads) cnt[k] = 0u;
    if (tid == 0) s_occ = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < P; i += kBinPlaceThreads) {
        const float4 r = tmp[bs + i];
        // (the record's key is computed again from its coordinates: the placeholder of a non-finite point fails the finiteness test)
        // (bin_count, bin_scatter and this kernel each derive the key from the coordinates with the same inlined expressions,
        // compiled without contraction: they agree.  Should they ever not, the clamp keeps the index inside cnt[] -- a misplaced
        // record instead of a write outside LDS; tests/test_gpu_index.py compares every product of this build with the atomic build's.)
        const uint32_t k = min(point_key(r.x, r.y, r.z, g, st, tg, tile_major, nkeys) - k0, K - 1u);
        const uint32_t a = atomicAdd(&cnt[k], 1u);
        rkey[i] = k | (a << 16);
        ridx[i] = __float_as_uint(r.w);
    }
    __syncthreads();
    // exclusive scan of the K counts (kBinKeysMax / kBinPlaceThreads consecutive keys per thread), occupied keys counted on the way
    constexpr uint32_t kPer = kBinKeysMax / kBinPlaceThreads;
    uint32_t v[kPer], sum = 0, nz = 0;
#pragma unroll
    for (uint32_t t = 0; t < kPer; ++t) {
        const uint32_t k = tid * kPer + t;
        v[t] = k < K ? cnt[k] : 0u;
        sum += v[t];
        nz += v[t] != 0u;
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t2 = __shfl_up(inc, o);
        if ((tid & 63) >= (unsigned)o) inc += t2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nz += __shfl_xor(nz, o);
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    if ((tid & 63) == 0 && nz) atomicAdd(&s_occ, nz);
    __syncthreads();
    uint32_t run = inc - sum;
    for (uint32_t w = 0; w < (tid >> 6); ++w) run += wsum[w];
#pragma unroll
    for (uint32_t t = 0; t < kPer; ++t) {
        const uint32_t k = tid * kPer + t;
        if (k < K) { cnt[k] = run; cs[k0 + k] = bs + run; }
        run += v[t];
    }
    if (bin == nbins - 1 && tid == 0) cs[nkeys + 1] = bs + P;              // the total behind the last key
    __syncthreads();
    // records by (start of their key + arrival): the members of a key sit together, in arrival order
    for (uint32_t i = tid; i < P; i += kBinPlaceThreads) slot[cnt[rkey[i] & 0xFFFFu] + (rkey[i] >> 16)] = ridx[i];
    __syncthreads();
    // rank inside the key = members with a smaller original index (keys hold a handful of points: a cell; P <= kBinCap bounds it)
    for (uint32_t i = tid; i < P; i += kBinPlaceThreads) {
        const uint32_t k = rkey[i] & 0xFFFFu, me = ridx[i];
        const uint32_t s0 = cnt[k], e0 = (k + 1 < K) ? cnt[k + 1] : P;
        uint32_t rank = 0;
        for (uint32_t j = s0; j < e0; ++j) rank += slot[j] < me ? 1u : 0u;
        rkey[i] = s0 + rank;
    }
    __syncthreads();
    for (uint32_t i = tid; i < P; i += kBinPlaceThreads) slot[rkey[i]] = i;
    __syncthreads();
    for (uint32_t t = tid; t < P; t += kBinPlaceThreads) pts[bs + t] = tmp[bs + slot[t]];        // final order, coalesced stores
    // occupied keys of the whole index: the last bin to arrive publishes the count (in front of the prefix sums: the normals kernel
    // picks its path by it; and, for the edge adaptation, to the pinned host block)
    // (count and ticket travel in ONE 64-bit atomic -- occupied keys in the high word, arrived bins in the low one: no ordering
    // between two atomics to enforce, no fence: an agent-scope fence per block, with the bin's 32 KB of fresh stores behind it, made
    // this kernel 34 us instead of 10)
    if (tid == 0) {
        const unsigned long long mine = ((unsigned long long)s_occ << 32) | 1ull;
        const unsigned long long old = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(occ_ticket), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)old == nbins - 1u) {
            const uint32_t occ = (uint32_t)(old >> 32) + s_occ;
            if (occ_out) __hip_atomic_store(occ_out, occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (occ_host) __hip_atomic_store(occ_host, (1ull << 32) | (unsigned long long)occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// cells up to this population are ranked by original index (quadratic in the population); TC_RANK_QUADRATIC_MAX overrides (read
// per call: the tests lower it to reach the strict-order re-sort with a cell of a few thousand points)
static uint32_t rank_quadratic_max() {
    const char *e = getenv("TC_RANK_QUADRATIC_MAX");
    return e ? (uint32_t)std::max(1, atoi(e)) : kRankQuadraticMax;
}

static bool binned_build_enabled() {          // TC_INDEX_BINNED=0: the atomic counting sort only (A/B; read per call: the tests flip it)
    const char *e = getenv("TC_INDEX_BINNED");
    return !(e && atoi(e) == 0);
}

tc_status build_index(tc_context *ctx, DeviceIndex &ix, const float *d_xyz, size_t n, float cell_factor,
                      const GridGeom *reuse_geom, const IcpState *d_state_transform, const TileGeom *tile_major,
                      float min_cell_edge, float target_ppo, bool strict_order) {
    if (n == 0 || n >= 0xFFFFFFF0ull) return fail(ctx, TC_INVALID_DATA, "build_index: bad point count");
    ctx->stat_indexed_points += n;
    ctx->stat_index_builds += 1;
    ix.vor_valid = false;
    ix.pts12_valid = false;
    ix.occ_host_valid = false;
    hipStream_t st = ctx->stream;
    const uint32_t n32 = (uint32_t)n;
    const int nb = (int)((n + 255) / 256);
    const int dbg = debug_flags();
    const uint32_t rank_max = rank_quadratic_max();

    if (reuse_geom) {
        ix.geom = *reuse_geom;
        ix.geom.n = n32;
    } else {
        float mn[3], mx[3], rmn[3], rmx[3];
        bool clamped = false;
        if (tc_status s = cloud_bbox_impl(ctx, d_xyz, n, mn, mx, rmn, rmx, &clamped)) return s;
        if (dbg & 2048) clamped = false, std::memcpy(rmn, mn, sizeof(mn)), std::memcpy(rmx, mx, sizeof(mx));   // TC_DEBUG & 2048: exact box only
        for (int c = 0; c < 3; ++c) {
            if (!(mn[c] <= mx[c]) || !std::isfinite(mn[c]) || !std::isfinite(mx[c])) { mn[c] = 0.0f; mx[c] = 0.0f; rmn[c] = 0.0f; rmx[c] = 0.0f; }
            ix.exact_min[c] = mn[c]; ix.exact_max[c] = mx[c];
        }
        derive_geom(ix.geom, rmn, rmx, n, cell_factor, min_cell_edge);
        ix.geom.clamped = clamped ? 1 : 0;
        if ((dbg & 256) && clamped)
            fprintf(stderr, "[tc] index: box clamped to %g..%g %g..%g %g..%g (exact %g..%g %g..%g %g..%g)\n", rmn[0], rmx[0], rmn[1], rmx[1],
                    rmn[2], rmx[2], mn[0], mx[0], mn[1], mx[1], mn[2], mx[2]);
    }
    // The volume-based edge assumes the cloud fills its box.  A surface (depth map, LiDAR sweep) or a
    // clustered cloud puts tens of points into every OCCUPIED cell.  The prefix sum counts the occupied
    // cells anyway: when there are more than twice the wanted points per occupied cell the edge is shrunk
    // (points per cell of a surface ~ h^2) and the index rebuilt, up to three times and 32 cells per point.
    // TC_DEBUG & 256 prints the decisions, & 512 disables the adaptation.
    // Only for clouds of >= 2^18 points: the check costs a host synchronisation at the end of the build
    // (~15 us of launch bubble), which a 24 k-point LiDAR frame pipeline feels (-14 % frames/s) and a
    // 1 M-point cloud does not (-0.5 %), while the gain scales with the cloud (TUM-shaped 1 M: 2-3x).
    // (TC_SURFACE_PPO_MULT: tuning experiments -- the points per occupied cell an adapted surface grid aims for, times this)
    if (const char *e = getenv("TC_SURFACE_PPO_MULT")) { const double m = atof(e); if (m > 0.0) target_ppo = (float)(target_ppo * m); }
    const bool adapt = target_ppo > 0.0f && !reuse_geom && !tile_major && n >= kAdaptMinPoints && !(dbg & 512);
    uint32_t nkeys_final = 0;
    const uint32_t *cs_final = nullptr;
    for (int attempt = 0;; ++attempt) {
        const GridGeom g = ix.geom;
        TileGeom tg{};
        uint32_t nkeys = g.ncell;          // number of counting-sort keys
        if (tile_major) {
            tg = *tile_major;
            ix.tile = tg;
            const uint64_t nk = (uint64_t)tg.ntiles * tg.cpt;
            if (nk >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "tile-major key space too large");
            nkeys = (uint32_t)nk;
        }

        if (tc_status s = ensure(ctx, ix.pts, (n + kPtsPad) * sizeof(float4))) return s;
        if (tc_status s = ensure(ctx, ix.cell_of, n * sizeof(uint32_t))) return s;
        if (tc_status s = ensure(ctx, ix.slot, n * sizeof(uint32_t))) return s;
        if (tc_status s = ensure(ctx, ix.arrival, n * sizeof(uint32_t))) return s;
        // nkeys cells + the bucket of the non-finite points behind them
        // (+ the oversized-cell flag; rounded up to 256 bytes: a memset whose size is not a multiple of its fill kernel's vector width is
        // TWO fill kernels on the stream, 3.6 + 4.4 us instead of one)
        const size_t fill_bytes = (((size_t)nkeys + 2) * sizeof(uint32_t) + 255) & ~(size_t)255;
        if (tc_status s = ensure(ctx, ix.fill, fill_bytes)) return s;
        if (tc_status s = ensure(ctx, ix.cell_start, (kCellStartFront + (size_t)nkeys + 2 + kCellStartPad) * sizeof(uint32_t))) return s;
        uint32_t *const cs = (uint32_t *)ix.cell_start.p + kCellStartFront;       // zeros in front (the ICP window of cell 0 starts at -1)
        nkeys_final = nkeys; cs_final = cs;
        const bool check = adapt && attempt < 3;
        volatile uint32_t *h_occ = (volatile uint32_t *)((char *)ctx->pinned + 2048 + 8192);      // [0] = occupied cells, [1] = written

        // ---- binned placement (see bin_count_kernel): dense-ish grids of large clouds, no global atomic per point ----
        bool binned_done = false;
        if (binned_build_enabled() && pinned_poll_enabled() && !strict_order && n >= kAdaptMinPoints && n <= (1u << 24) && nkeys < 0x7FFFFFF0u) {
            const uint32_t keys = nkeys + 1u;               // + the bucket of the non-finite points
            uint32_t nbins = std::max<uint32_t>((n32 + kBinTarget - 1) / kBinTarget, (keys + kBinKeysMax - 1) / kBinKeysMax);
            const uint32_t kpb = (keys + nbins - 1) / nbins;           // keys per bin
            nbins = (keys + kpb - 1) / kpb;
            if (nbins <= kBinMaxBins && kpb <= kBinKeysMax) {
                // scratch in ix.fill: {occupied cells, arrived bins} as one 8-byte word (+ 8 bytes: the rows below are read 16 bytes at a time)
                // | [nbins][kBinBlocks] counts / offsets | nbins totals | nbins + 1 starts
                const size_t words = 4 + (size_t)kBinBlocks * nbins + nbins + (nbins + 1);
                if (tc_status s = ensure(ctx, ix.fill, std::max(fill_bytes, words * sizeof(uint32_t)))) return s;
                uint32_t *occ_ticket = (uint32_t *)ix.fill.p, *cnt = occ_ticket + 4, *tot = cnt + (size_t)kBinBlocks * nbins, *binstart = tot + nbins;
                volatile uint32_t *h_max = (volatile uint32_t *)((char *)ctx->pinned + 2048 + 8192 + 256);      // [0] = largest bin, [1] = written
                unsigned long long *d_max = (unsigned long long *)pinned_dev_ptr(ctx, (const void *)h_max);
                if (!d_max) return fail(ctx, TC_GPU, "index build: the pinned block has no device address");
                h_max[0] = 0u; h_max[1] = 0u;
                {
                    ProfScope ps(ctx, "cell_bin_count");
                    hipLaunchKernelGGL(bin_count_kernel, dim3(kBinBlocks), dim3(kBinWalkThreads), 0, st, d_xyz, n32, g, d_state_transform, tg, tile_major ? 1 : 0, nkeys, kpb,
                                       nbins, cnt, reinterpret_cast<uint32_t *>((float4 *)ix.pts.p + n), (uint32_t *)ix.cell_start.p, cs + nkeys + 2);
                }
                {
                    ProfScope ps(ctx, "cell_bin_offsets");
                    hipLaunchKernelGGL(bin_offsets_kernel, dim3((nbins + 3) / 4), dim3(256), 0, st, cnt, nbins, tot);
                }
                if (tc_status s = ensure(ctx, ctx->build_tmp, (n + kPtsPad) * sizeof(float4))) return s;
                // the scatter is valid whatever the bins' populations are (a build that falls back has wasted these 10 us)
                {
                    ProfScope ps(ctx, "cell_bin_scatter");
                    hipLaunchKernelGGL(bin_scatter_kernel, dim3(kBinBlocks), dim3(kBinWalkThreads), 0, st, d_xyz, n32, g, d_state_transform, tg, tile_major ? 1 : 0, nkeys,
                                       kpb, nbins, (const uint32_t *)cnt, (const uint32_t *)tot, binstart, occ_ticket, d_max, (float4 *)ctx->build_tmp.p);
                }
                if (check) { h_occ[0] = 0u; h_occ[1] = 0u; }
                unsigned long long *d_occ_host = check ? (unsigned long long *)pinned_dev_ptr(ctx, (const void *)h_occ) : nullptr;
                {
                    ProfScope ps(ctx, "cell_bin_place");
                    hipLaunchKernelGGL(bin_place_kernel, dim3(nbins), dim3(kBinPlaceThreads), 0, st, (const float4 *)ctx->build_tmp.p, (const uint32_t *)binstart,
                                       nbins, kpb, nkeys, g, d_state_transform, tg, tile_major ? 1 : 0, cs, (float4 *)ix.pts.p, occ_ticket,
                                       (uint32_t *)ix.cell_start.p, d_occ_host);
                }
                TC_HIP_TRY(ctx, hipGetLastError());
                // (the placement runs -- or has found out that it must not -- while the host waits for the same number)
                if (tc_status s = wait_pinned_word(ctx, h_max + 1, "index build (bin populations)")) return s;
                if (h_max[0] <= kBinCap) {
                    binned_done = true;
                } else if (dbg & 256) {
                    fprintf(stderr, "[tc] index: largest bin holds %u points (> %u): atomic counting sort\n", h_max[0], kBinCap);
                }
            }
        }
        if (!binned_done) {
        TC_HIP_TRY(ctx, hipMemsetAsync(ix.fill.p, 0, fill_bytes, st));
        // records past the end: huge finite coordinates -> d2 = +inf, never a match (kernels may read, never select them)
        {
            ProfScope ps(ctx, "cell_hist");
            hipLaunchKernelGGL(cell_hist_kernel, dim3(nb), dim3(256), 0, st, d_xyz, n32, g, d_state_transform, tg, tile_major ? 1 : 0, nkeys,
                               (uint32_t *)ix.cell_of.p, (uint32_t *)ix.fill.p, (uint32_t *)ix.arrival.p,
                               reinterpret_cast<uint32_t *>((float4 *)ix.pts.p + n), (uint32_t *)ix.cell_start.p, cs + nkeys + 2, rank_max);
        }
        {
            ProfScope ps(ctx, "cell_scan");
            // (the number of occupied cells also goes to the first word in front of the prefix sums: view.cell_start[-kCellStartFront])
            // (with the edge adaptation on, the count also goes to the pinned host block: polled below while place / rerank run)
            if (check) { h_occ[0] = 0u; h_occ[1] = 0u; }
            unsigned long long *d_occ_host = (check && pinned_poll_enabled()) ? (unsigned long long *)pinned_dev_ptr(ctx, (const void *)h_occ) : nullptr;
            if (check && pinned_poll_enabled() && !d_occ_host) return fail(ctx, TC_GPU, "index build: the pinned block has no device address");
            if (tc_status s = exclusive_scan_u32(ctx, (const uint32_t *)ix.fill.p, nkeys + 1, cs, ix.blocksum, (uint32_t *)ix.cell_start.p, d_occ_host)) return s;
        }
        if (tc_status s = ensure(ctx, ctx->build_tmp, (n + kPtsPad) * sizeof(float4))) return s;       // (shared by every build of the context)
        {
            ProfScope ps(ctx, "cell_place");
            hipLaunchKernelGGL(place_kernel, dim3(nb), dim3(256), 0, st, d_xyz, n32, nkeys, (const uint32_t *)ix.cell_of.p, (const uint32_t *)cs,
                               (const uint32_t *)ix.arrival.p, (float4 *)ctx->build_tmp.p);
        }
        {
            ProfScope ps(ctx, "cell_rerank");
            hipLaunchKernelGGL(rerank_kernel, dim3(nb), dim3(256), 0, st, n32, g, d_state_transform, tg, tile_major ? 1 : 0, nkeys, (const uint32_t *)cs,
                               (const float4 *)ctx->build_tmp.p, (float4 *)ix.pts.p, rank_max);
        }
        }
        TC_HIP_TRY(ctx, hipGetLastError());
        if (!check) break;
        // the rest of the build is already enqueued (the usual outcome is to keep it) and keeps running while the host waits for
        // the scan's word only: the caller's next kernel is enqueued under place / rerank, not behind a drained stream
        if (pinned_poll_enabled()) {
            if (tc_status s = wait_pinned_word(ctx, h_occ + 1, "index build (occupied cells)")) return s;
        } else {
            const uint32_t nscan = (nkeys + 1 + kScanTile - 1) / kScanTile;
            TC_HIP_TRY(ctx, hipMemcpyAsync((void *)h_occ, (const uint32_t *)ix.blocksum.p + 2 * nscan, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            TC_HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        const uint32_t occ_now = h_occ[0];
        const double ppo = (double)n / (double)std::max<uint32_t>(occ_now, 1u);
        ix.occ_host = occ_now; ix.occ_host_valid = true;        // (of THIS attempt's grid: reset below when another attempt follows)
        if (dbg & 256)
            fprintf(stderr, "[tc] index: n %zu h %.5f grid %d x %d x %d = %u cells (%.2f n), %u occupied, %.2f points each (want %.1f)\n", n,
                    g.h, g.gx, g.gy, g.gz, g.ncell, (double)g.ncell / (double)n, occ_now, ppo, target_ppo);
        if (!(ppo > 2.0 * target_ppo)) break;
        double h = (double)g.h * std::min(0.8, std::max(0.35, std::sqrt((double)target_ppo / ppo)));
        if (min_cell_edge > 0.0f) h = std::max(h, (double)min_cell_edge);
        GridGeom ng = g;
        set_cell_edge(ng, h, n, 32.0);      // measured on a 1 M-point depth-map surface: 16 -> 32 cells per point -8 % normals, -10 % ICP; 64: no further gain
        if (!(ng.h < 0.95f * g.h)) break;                     // budget or minimum edge reached
        ix.geom = ng;
        ix.occ_host_valid = false;
    }
    if (strict_order) {
        // Ranks that split the cell-sorted order between them (TC_SHARD_SPATIAL, sharded normals) need the SAME order on every
        // rank, also inside a cell of more than kRankQuadraticMax (2^20) points, where rerank_kernel keeps the atomic arrival order:
        // one host round trip for the flag, and -- only then -- a stable LSD radix sort of (cell, original index) replaces the
        // order (rocPRIM, the library primitive the voxel filter's sort path already uses), records gathered again.
        uint32_t *h_big = (uint32_t *)((char *)ctx->pinned + 2048 + 8192 + 64);
        TC_HIP_TRY(ctx, hipMemcpyAsync(h_big, (const uint32_t *)ix.fill.p + nkeys_final + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        TC_HIP_TRY(ctx, hipStreamSynchronize(st));
        if (*h_big) {
            if (tc_status s = ensure(ctx, ctx->overflow, n * sizeof(uint32_t))) return s;
            uint32_t *iota = (uint32_t *)ix.arrival.p, *keys_out = (uint32_t *)ctx->overflow.p;
            hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, st, iota, n32);
            unsigned bits = 1;
            while (bits < 32 && (1ull << bits) <= (unsigned long long)nkeys_final) ++bits;
            size_t temp_bytes = 0;
            TC_HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint32_t *)ix.cell_of.p, keys_out, (const uint32_t *)iota,
                                                      (uint32_t *)ix.slot.p, n, 0u, bits, st));
            if (tc_status s = ensure(ctx, ix.blocksum, temp_bytes)) return s;
            TC_HIP_TRY(ctx, rocprim::radix_sort_pairs(ix.blocksum.p, temp_bytes, (const uint32_t *)ix.cell_of.p, keys_out, (const uint32_t *)iota,
                                                      (uint32_t *)ix.slot.p, n, 0u, bits, st));
            ProfScope ps(ctx, "cell_rank_gather_strict");
            hipLaunchKernelGGL(rank_gather_kernel, dim3(nb), dim3(256), 0, st, d_xyz, n32, nkeys_final, (const uint32_t *)ix.cell_of.p,
                               (const uint32_t *)cs_final, (const uint32_t *)ix.slot.p, (float4 *)ix.pts.p, 1, rank_max);
            TC_HIP_TRY(ctx, hipGetLastError());
        }
    }
    return TC_OK;
}

tc_status gather_normals(tc_context *ctx, DeviceIndex &ix, const float *d_normals, size_t stride) {
    const size_t n = ix.geom.n;
    if (tc_status s = ensure(ctx, ix.normals, n * sizeof(float4))) return s;
    ProfScope ps(ctx, "gather_normals");
    hipLaunchKernelGGL(gather_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float4 *)ix.pts.p, (uint32_t)n, d_normals, stride, (float4 *)ix.normals.p);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

}  // namespace tc
