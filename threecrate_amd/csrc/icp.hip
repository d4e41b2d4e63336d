// icp.hip -- HK3 icp_correspond_reduce + device-side solve for point-to-point and
// point-to-plane ICP (threecrate-algorithms/src/registration.rs:258-370, :508-602).
//
// One iteration = two launches on the context's stream, no host round trip:
//   icp_correspond_reduce_kernel : per source point (visited in target-cell order):
//        ts = current_transform * s            (registration.rs:285-289 / :540-544, same f32 formula)
//        exact 1-NN of ts in the target grid   (:87-107; ring search with an exact stopping rule
//                                               that also covers queries outside the grid)
//        write the matched target index        (ICPResult.correspondences, :22-23)
//        accumulate the packed normal equations of the iteration in f64:
//          p2plane: 21 upper-tri AtA + 6 Atb + sum b^2 + count = 29 words (:409-428, :453-471)
//          p2point: sum s, sum q, sum s q^T, sum |s-q|^2, count = 17 words (:154-172, :206-218)
//        wave64 shuffle reduction -> LDS -> one row of per-block partials (fixed order,
//        no float atomics: results are bit-reproducible run to run).
//   icp_finalize_kernel : fixed-order sum of the partials, 6x6 Cholesky/LU (p2plane) or 3x3
//        Kabsch via one-sided Jacobi SVD (p2point) in f64, compose `delta * current`, mse,
//        |prev_mse - mse| < threshold test, all state kept in an IcpState in HBM.
// Algorithmic HBM bytes per source point and iteration: 12 src + 12 matched target
// + 12 matched normal + 4 index out = 40 B (p2plane), 28 B (p2point).
#include "tc_internal.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>

namespace tc {

__device__ __forceinline__ void iso_apply(const float q[4], const float t[3], float x, float y, float z,
                                          float &ox, float &oy, float &oz) {
    // nalgebra: t2 = (qv x p) * 2; p' = (t2 * w + qv x t2) + p; then + translation
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

// The first 64 bytes of the IcpState (rotation, translation, mse words, iteration count, flags, max_dist) in ONE scalar load
// issued before anything else: a kernel that tests `done`, then fetches the transform, then the distance cut starts with three
// dependent round trips to memory (~0.5 us each), and every kernel of an iteration starts like that.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
struct IcpHeader {
    float q[4], t[3];
    float prev_mse, conv_thr, max_dist;
    uint32_t iterations;
    int32_t done;
    float d_ang, d_t;          // (words 17 .. 19: a second scalar load, issued with the first)
    uint32_t d_run;
};
static_assert(offsetof(IcpState, done) == 48 && offsetof(IcpState, max_dist) == 60 && offsetof(IcpState, iterations) == 36 &&
              offsetof(IcpState, prev_mse) == 28 && offsetof(IcpState, conv_thr) == 56, "IcpState header layout");
__device__ __forceinline__ IcpHeader load_header(const IcpState *st) {
    const u32x16 h = *reinterpret_cast<const u32x16 *>(st);
    IcpHeader o;
    o.q[0] = __uint_as_float(h[0]); o.q[1] = __uint_as_float(h[1]); o.q[2] = __uint_as_float(h[2]); o.q[3] = __uint_as_float(h[3]);
    o.t[0] = __uint_as_float(h[4]); o.t[1] = __uint_as_float(h[5]); o.t[2] = __uint_as_float(h[6]);
    o.prev_mse = __uint_as_float(h[7]);
    o.iterations = h[9];
    o.done = (int32_t)h[12];
    o.conv_thr = __uint_as_float(h[14]);
    o.max_dist = __uint_as_float(h[15]);
    o.d_ang = st->d_ang; o.d_t = st->d_t; o.d_run = st->d_run;
    return o;
}

__device__ __forceinline__ uint32_t xcd_remap_icp(uint32_t b, uint32_t nb) {
    const uint32_t per = nb >> 3;          // nb is a multiple of 8
    return (b & 7u) * per + (b >> 3);
}

// squared distance from a query to its projection onto the grid box, shaved by a safety factor
// (the bound needs every record inside the box: with a clamped box -- `ext` -- it is dropped)
__device__ __forceinline__ float outside_d2(float x, float y, float z, float qx, float qy, float qz, int ext) {
    const float ex = x - qx, ey = y - qy, ez = z - qz;
    return ext ? 0.0f : (ex * ex + ey * ey + ez * ez) * 0.9999f;
}

// squared distance from q to the box of cell index c along one axis, shaved by the cell-assignment
// fuzz (conservative: never larger than the true gap)
// `ext`: the grid's box is clamped (GridGeom::clamped): the first / last cell of the axis also holds the records
// beyond the box, so it has no face on that side
__device__ __forceinline__ float axis_gap(float q, float mn, float h, int c, int last, int ext) {
    const float lo = mn + (float)c * h, hi = lo + h;
    float a = lo - q, b = q - hi;
    if (ext) {
        a = (c == 0) ? -INFINITY : a;
        b = (c == last) ? -INFINITY : b;
    }
    return fmaxf(fmaxf(a, b) - 2e-3f * h, 0.0f);
}

// Sum NACC (<= 32) per-lane values over the 64 lanes of a wave into row[0 .. NACC) (f64, += by one lane per slot).
// A transposing reduction: at every step a lane keeps one half of its values and sends the other half to its partner, so
// the work halves with the lane distance (32 -> 16 -> 8 -> 4 -> 2 values per lane; 3 instructions per kept value) instead
// of six full butterfly steps per value; lane j of row 0 ends with the totals of slots s(j) and s(j) + 16.  The DPP
// partners are i^1, i^2, i^7 (row_half_mirror) and i^15 (row_mirror); the keep / send choice of a step must agree between
// the partners of all LATER steps, hence f1 = b0^b2, f2 = b1^b2, f3 = b2^b3, f4 = b3.  Fixed tree: reproducible sums.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int N, int CTRL>
__device__ __forceinline__ void transpose_step(const float (&in)[2 * N], float (&out)[N], bool f) {
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const float keep = f ? in[2 * m + 1] : in[2 * m], send = f ? in[2 * m] : in[2 * m + 1];
        out[m] = keep + dpp_move<CTRL>(send);
    }
}
template <int NACC>
__device__ __forceinline__ void wave_fold_transposed(const float (&acc)[NACC], double *__restrict__ row, int lane) {
    static_assert(NACC <= 32, "32 slots");
    float v[32], w[16], u[8], t4[4], t2[2];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = i < NACC ? acc[i] : 0.0f;
    const int b = lane;
    const bool f1 = ((b ^ (b >> 2)) & 1) != 0, f2 = (((b >> 1) ^ (b >> 2)) & 1) != 0, f3 = (((b >> 2) ^ (b >> 3)) & 1) != 0,
               f4 = ((b >> 3) & 1) != 0;
    transpose_step<16, 0xB1>(v, w, f1);       // quad_perm [1,0,3,2]
    transpose_step<8, 0x4E>(w, u, f2);        // quad_perm [2,3,0,1]
    transpose_step<4, 0x141>(u, t4, f3);      // row_half_mirror
    transpose_step<2, 0x140>(t4, t2, f4);     // row_mirror
#pragma unroll
    for (int m = 0; m < 2; ++m) {             // the four rows hold the same slots lane by lane
        t2[m] += __shfl_xor(t2[m], 16);
        t2[m] += __shfl_xor(t2[m], 32);
    }
    if (lane < 16) {
        const int s0 = (f4 ? 8 : 0) + (f3 ? 4 : 0) + (f2 ? 2 : 0) + (f1 ? 1 : 0);
        if (s0 < NACC) row[s0] += (double)t2[0];
        if (s0 + 16 < NACC) row[s0 + 16] += (double)t2[1];
    }
}

template <int NACC, int NW = kIcpBlock / 64>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], double *__restrict__ out_row, double (*sm)[TC_ICP_SUMS_STRIDE],
                                                   double extra = 0.0) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        acc[i] = v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) sm[w][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < TC_ICP_SUMS_STRIDE) {
        double s = 0.0;
        if (threadIdx.x < NACC) {
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) s += sm[w2][threadIdx.x];
        }
        out_row[threadIdx.x] = s + extra;
    }
}

// ---- correspondence search + reduction ---------------------------------------------------------
// One lane per source point, points visited in target-cell (tile-major) order so that the lanes
// of a wave read neighbouring rows of the cell-sorted target (L1 / L2 hits, merged requests).
//
// Warm start (exact): the previous iteration's match p of a source point is a real target point,
// so ub = |T s - p|^2 bounds the new nearest-neighbour distance from above.  Only the rows / cells
// of the 3x3x3 block that intersect that ball can hold the answer; everything else is skipped.
// In steady state that leaves ~1-3 short spans (~5 candidates) per query instead of 9 (~40).
//
// Candidate loop: the surviving spans live in registers; one flattened, software-pipelined loop
// walks them (the record of step i+1 is requested before step i is evaluated) so that a lane
// with few candidates does not wait on a row-by-row schedule.  The running best is the minimum
// of the 64-bit key (d2 bits, position): independent of visiting order, ties -> lowest position.
//
// Sums: per-pair terms and products are f32 exactly as the reference forms them
// (registration.rs:417-427); a lane adds its handful of pairs in f32, then wave shuffles -> LDS
// -> per-block partial rows in f64, folded in a fixed order by icp_finalize_kernel.

// The (up to) nine row spans of a query live in LDS ([row][lane], one ds_read_b64 per span switch)
// instead of 18 registers + a select chain: the main kernel is bound by loads in flight, i.e. by
// waves per SIMD, i.e. by its VGPR count.
constexpr int kSpanRows = 9;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// squared distance of record c to the query, same operations and order as d2_nc
// (dx*dx + dy*dy + dz*dz, no FMA), with the (x, y) part as packed f32
__device__ __forceinline__ float d2_packed(const f32x3 &c, const f32x2 &qxy, float qz) {
    const f32x2 dxy = c.xy - qxy;
    const float dz = c.z - qz;
    const f32x2 sxy = dxy * dxy;
    return (sxy.x + sxy.y) + dz * dz;
}

// raw buffer descriptor over a whole allocation (no range check: 4 GiB window): buffer loads take a
// 32-bit byte offset per lane (no 64-bit address arithmetic) and accept dword-aligned 16-byte reads
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xFFFFFFFF, 0x00020000);
}

// exact 1-NN of (x, y, z); ub2 = a valid upper bound of the squared NN distance (or +inf)
// -DTC_PHASE_STAMPS builds (tools/dev/build_variant.sh): wave 0 of every main-pass block adds up the shader clock (s_memtime)
// it spends in each phase; TC_DEBUG & 1024 prints the means.  Nothing of it exists in a normal build.
#ifdef TC_PHASE_STAMPS
#define TC_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tl; tl = t_; } while (0)
#define TC_STAMP_ARGS , unsigned long long (&ph)[8], unsigned long long &tl
#define TC_STAMP_PASS , ph, tl
#else
#define TC_STAMP(i) do { } while (0)
#define TC_STAMP_ARGS
#define TC_STAMP_PASS
#endif
#ifndef TC_ICP_STEP
#define TC_ICP_STEP 4          // records per candidate step
#endif
// track2 (wave-uniform): the candidate loop also keeps the SECOND smallest distance, and low2 returns a lower bound of the squared
// distance from the query to every target point other than the one found (the second-neighbour certificate of the main pass).
template <bool STATS = false, bool T2 = false>
__device__ __forceinline__ void nn_search_pruned(const GridView &gv, float x, float y, float z, float ub2,
                                                 float &best, uint32_t &bestj, bool &refine, float max_dist,
                                                 uint2 (*spans)[kIcpBlock], uint32_t &nsteps, const float *pts12, bool track2, float &low2 TC_STAMP_ARGS
                                                 ) {
    const GridGeom &g = gv.g;
    int cx, cy, cz;
    float out2;
    const float qx = fminf(fmaxf(x, g.minx), g.maxx), qy = fminf(fmaxf(y, g.miny), g.maxy),
                qz = fminf(fmaxf(z, g.minz), g.maxz);
    cx = cell_coord(qx, g.minx, g.inv_h, g.gx);
    cy = cell_coord(qy, g.miny, g.inv_h, g.gy);
    cz = cell_coord(qz, g.minz, g.inv_h, g.gz);
    const float fx = (qx - g.minx) * g.inv_h - (float)cx, fy = (qy - g.miny) * g.inv_h - (float)cy,
                fz = (qz - g.minz) * g.inv_h - (float)cz;
    out2 = outside_d2(x, y, z, qx, qy, qz, g.clamped);
    // squared distance (shaved by the cell-assignment fuzz) from q' to the neighbouring slabs
    const float lo_x = fmaxf(fx - 2e-3f, 0.0f) * g.h, hi_x = fmaxf(1.0f - fx - 2e-3f, 0.0f) * g.h;
    const float lo_y = fmaxf(fy - 2e-3f, 0.0f) * g.h, hi_y = fmaxf(1.0f - fy - 2e-3f, 0.0f) * g.h;
    const float lo_z = fmaxf(fz - 2e-3f, 0.0f) * g.h, hi_z = fmaxf(1.0f - fz - 2e-3f, 0.0f) * g.h;
    const float ax2[3] = {lo_x * lo_x, 0.0f, hi_x * hi_x};
    const float ay2[3] = {lo_y * lo_y, 0.0f, hi_y * hi_y};
    const float az2[3] = {lo_z * lo_z, 0.0f, hi_z * hi_z};
    const float ub = ub2 - out2;     // budget left inside the box (|p-q|^2 >= |p-q'|^2 + |q-q'|^2)
    // One unaligned 16-byte window of cell_start per row, [row + cx - 1, row + cx + 2], issued for all
    // nine rows back to back (one round trip): it holds the start of the left, own and right cell and
    // the end of the right cell.  (cell_start is padded by kCellStartPad entries.)
    const __amdgpu_buffer_rsrc_t cs_rsrc = raw_rsrc(gv.cell_start - 1);      // window of cell c starts at entry c - 1 (zeros in front of the array)
    // per-axis facts shared by the nine rows; row starts by adding uniform strides to the centre row
    // (no per-row integer multiplies: v_mul_lo_u32 is quarter rate)
    const bool oky[3] = {cy > 0, true, cy < g.gy - 1}, okz[3] = {cz > 0, true, cz < g.gz - 1};
    const bool has_l = cx > 0, has_r = cx < g.gx - 1;
    // (opaque to the optimiser: it otherwise re-associates every row back into ((cz + dz) * gy + cy + dy) * gx)
    const int stride_y = __builtin_amdgcn_readfirstlane(g.gx), stride_z = __builtin_amdgcn_readfirstlane(g.gx * g.gy);
    const int row_c = (cz * g.gy + cy) * g.gx + cx;        // window start (+1) of the centre row
    uint32_t s0[kSpanRows], e0[kSpanRows];
    uint32_t mask = 0;
    // (the budget left for the row part once the left / right cell's x distance is paid: one subtraction each instead of an addition
    // per row and side; the centre terms of ay2 / az2 are zeros that are not added -- 23 instructions less per search.  Which side of
    // an exact boundary a rounding lands on is immaterial: the distances are shaved by 2e-3 h, thousands of ulps)
    const float ub_l = ub - ax2[0], ub_r = ub - ax2[2];
#pragma unroll
    for (int k = 0; k < kSpanRows; ++k) {
        const int dz = k / 3 - 1, dy = k % 3 - 1;
        const float r2 = dy == 0 ? az2[dz + 1] : dz == 0 ? ay2[dy + 1] : ay2[dy + 1] + az2[dz + 1];
        const bool on = oky[dy + 1] && okz[dz + 1] && !(r2 > ub);
        // x window: the left / right cell only if the ball reaches it
        const bool left = has_l && !(r2 > ub_l), right = has_r && !(r2 > ub_r);
        const int row = row_c + dz * stride_z + dy * stride_y;
        // (a lane whose ball does not reach the row takes no part in the read: the pass is bound by what goes through the
        // texture path as much as by instructions -- 44.2 -> 41.2 us per pass against reading a dummy window branch-free.  Measured
        // on top of it and lost: two 4-byte reads instead of the window (42.5), an 8-byte read for the lanes whose ball stays
        // inside their own column of cells (47.0), masking the 2nd .. 4th record read of a candidate step (46.4): extra exec
        // regions cost more than the bytes they save)
        u32x4 w = {0u, 0u, 0u, 0u};
        if (on) w = __builtin_amdgcn_raw_buffer_load_b128(cs_rsrc, (uint32_t)row << 2, 0, 0);
        // window = starts of the cells cx-1 .. cx+2; span = [start of cell cx - left, start of cell cx + right + 1)
        s0[k] = left ? w.x : w.y;
        e0[k] = right ? w.w : w.z;
        if (on && s0[k] != e0[k]) mask |= 1u << k;         // rows the ball reaches, non-empty spans only
    }
#pragma unroll
    for (int k = 0; k < kSpanRows; ++k) spans[k][threadIdx.x] = make_uint2(s0[k], e0[k]);
    TC_STAMP(1);
    // Flattened walk over the surviving spans, four records per step (four independent gathers in
    // flight per lane); a lane switches to its next span as soon as the current one is exhausted.
    // A step may read up to three records past its span: real target points of the next cells (or
    // the +inf padding behind the array), harmless as extra candidates.  Rows are visited in ascending
    // order = ascending position, so the strict '<' keeps the lowest position among equal distances.
    const __amdgpu_buffer_rsrc_t pt_rsrc = raw_rsrc(pts12);
    // the query as the three register PAIRS the twelve words of a step pair up with (see the loop): (x, y), (z, x), (y, z)
    const f32x2 qxy = {x, y}, qzx = {z, x}, qyz = {y, z};
    best = INFINITY;
    bestj = 0xFFFFFFFFu;
    // The span change is a select inside ONE divergent loop -- a lane leaves when its last span ends; the next span is fetched
    // from LDS at the top of every step -- instead of two nested exec-mask regions per step (`if (j >= e) { if (!mask) break; .. }`):
    // the loop is a single basic block of 68 instructions, 40.1 -> 39.1 us per pass on average (moving 46.0 -> 44.9, aligned 15.6 -> 15.3).
    // (Fully branch-free -- finished lanes kept reading the +inf records behind the array until the whole wave was done -- it was
    // 42.1 us: the pass is bound by what goes through the texture path as much as by instructions.)
    uint32_t j = 0, e = 0;
    {
        const int k0 = mask ? __ffs(mask) - 1 : 0;
        const uint2 se = spans[k0][threadIdx.x];
        j = mask ? se.x : 0u; e = mask ? se.y : 0u;
        mask &= mask - 1u;
    }
    float second = INFINITY;          // (track2) the second smallest distance seen, a record met twice counting twice (the safe side)
    // The candidate step, ONE text for both loops (a macro, not a lambda: wrapped in a generic lambda the plain loop came out one
    // instruction longer -- v_cmp_ne + select where the inline text gets v_add_co's carry for `mask != 0` -- and the benchmark's main pass
    // 1 % slower, profiles/r06_dense_trips_certificate.txt).  The comments of the step:
    //   * The candidates come from the PACKED copy of the sorted records (12 bytes each, DeviceIndex::pts12): four records = 48 bytes =
    //     THREE 16-byte reads (dword aligned) instead of four 12-byte reads of the 16-byte records (round 5, profiles/r05_ab_pack12.txt:
    //     42.9 -> 40.5 us per moving-phase pass, the cold first pass 57 -> 51 us, whole job +2.9 %).  Same bits.
    //   * 12 j by two instructions (v_mul_lo_u32 is quarter rate, and the compiler re-forms it from shifts); j < 2^28.
    //   * four records = three 16-byte registers -> the smallest of their four distances and its position (lowest on ties).
    //   * SECOND (the certificate's loop only): the step's own second smallest (of v0 .. v3 the loser of the final, or the smaller loser of
    //     the semi-finals), then the two smallest of {best, second, m, that}: ~8 instructions, as v_min / v_max instructions (no
    //     canonicalising v_max x, x).
#if TC_ICP_STEP == 8
#define TC_ICP_STEP8_LOADS \
        const f32x4 rd = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pt_rsrc, o + 48u, 0, 0)); \
        const f32x4 re = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pt_rsrc, o + 64u, 0, 0)); \
        const f32x4 rf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pt_rsrc, o + 80u, 0, 0));
#define TC_ICP_STEP8_QUAD { float m2; uint32_t im2; quad(rd, re, rf, j + 4, m2, im2); const bool b2 = m2 < m; m = b2 ? m2 : m; im = b2 ? im2 : im; }
#else
#define TC_ICP_STEP8_LOADS
#define TC_ICP_STEP8_QUAD
#endif
#define TC_ICP_SECOND_NONE
#define TC_ICP_SECOND_TRACK \
            auto mn = [](float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }; \
            auto mx = [](float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }; \
            const float m2 = mn(mx(mn(w0, w1), mn(w2, w3)), mn(mx(w0, w1), mx(w2, w3))); \
            second = mn(mx(best, m), mn(second, m2));
#define TC_ICP_CANDIDATE_STEP(SECOND) { \
        const int kn = mask ? __ffs(mask) - 1 : 0; \
        const uint2 nse = spans[kn][threadIdx.x]; \
        uint32_t o; \
        asm("v_lshlrev_b32 %0, 2, %1\n\tv_lshl_add_u32 %0, %1, 3, %0" : "=&v"(o) : "v"(j)); \
        auto add = [](float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }; \
        float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f; \
        auto quad = [&](const f32x4 &ra, const f32x4 &rb, const f32x4 &rc, uint32_t jb, float &m, uint32_t &im) { \
            const f32x2 d0 = ra.xy - qxy, d1 = ra.zw - qzx, d2 = rb.xy - qyz, d3 = rb.zw - qxy, d4 = rc.xy - qzx, d5 = rc.zw - qyz; \
            const f32x2 s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3, s4 = d4 * d4, s5 = d5 * d5; \
            const float v0 = add(add(s0.x, s0.y), s1.x), v1 = add(add(s1.y, s2.x), s2.y); \
            const float v2 = add(add(s3.x, s3.y), s4.x), v3 = add(add(s4.y, s5.x), s5.y); \
            const bool b01 = v1 < v0, b23 = v3 < v2; \
            const float m01 = b01 ? v1 : v0, m23 = b23 ? v3 : v2; \
            const uint32_t i01 = b01 ? jb + 1 : jb, i23 = b23 ? jb + 3 : jb + 2; \
            const bool bb = m23 < m01; \
            m = bb ? m23 : m01; \
            im = bb ? i23 : i01; \
            w0 = v0; w1 = v1; w2 = v2; w3 = v3; \
        }; \
        const f32x4 ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pt_rsrc, o, 0, 0)); \
        const f32x4 rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pt_rsrc, o + 16u, 0, 0)); \
        const f32x4 rc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pt_rsrc, o + 32u, 0, 0)); \
        TC_ICP_STEP8_LOADS \
        float m; uint32_t im; \
        quad(ra, rb, rc, j, m, im); \
        TC_ICP_STEP8_QUAD \
        SECOND \
        const bool upd = m < best; \
        best = upd ? m : best; \
        bestj = upd ? im : bestj; \
        j += TC_ICP_STEP; \
        if constexpr (STATS) ++nsteps; \
        const bool adv = j >= e && mask != 0u; \
        j = adv ? nse.x : j; \
        e = adv ? nse.y : e; \
        mask = adv ? (mask & (mask - 1u)) : mask; \
    }
    if constexpr (T2) {
        // (T2: only the certificate's instantiation of the main pass holds the second loop at all -- compiled into the plain kernel, the
        // inactive branch and its registers cost the benchmark pair 1.5 % of the whole job)
        if (track2) { while (j < e) TC_ICP_CANDIDATE_STEP(TC_ICP_SECOND_TRACK) }
        else { while (j < e) TC_ICP_CANDIDATE_STEP(TC_ICP_SECOND_NONE) }
    } else {
        while (j < e) TC_ICP_CANDIDATE_STEP(TC_ICP_SECOND_NONE)
    }
#undef TC_ICP_CANDIDATE_STEP
#undef TC_ICP_SECOND_TRACK
#undef TC_ICP_SECOND_NONE
#undef TC_ICP_STEP8_LOADS
#undef TC_ICP_STEP8_QUAD
    TC_STAMP(2);
    // Ring-1 exactness rule: best <= ((1 + m_f - 2e-3) h)^2 + |q - q'|^2, m_f >= 0 the clearance of q' to the block's faces.  Nearly
    // every lane passes it with m_f = 0 already (the nearest neighbour lies within one cell edge): only the others form the
    // clearance -- a real branch, skipped by the whole wave in most trips (~35 instructions of selects and minima per trip).
    const float h0 = (1.0f - 2e-3f) * g.h;
    // (track2) every target point other than the one found is at least this far (squared): the second smallest distance scanned; a
    // point in a pruned row or cell lies beyond the budget ub2; a point outside ring 1 beyond a cell edge (the bound of the rule below)
    low2 = (T2 && track2) ? fminf(fminf(second, ub2), h0 * h0 + out2) : 0.0f;
    refine = false;
    if (!(best <= h0 * h0 + out2) && !(ub2 < 0.0f)) {          // (ub2 < 0: a lane that does not search -- its flag is ignored anyway)
        // clearance of q' to the ring-1 block's faces, in cell edges: only a face with cells BEYOND it counts (where the grid ends at
        // or before the block's face nothing unscanned lies on that side -- a query outside the target's box sits ON the box face:
        // counting that face made its clearance 0 and sent most of the 6 % of the benchmark's source points that are outside the
        // target's box to the refine pass)
        const float big = 1.0e30f;
        const float lx = cx >= 2 ? fx : big, hx = cx <= g.gx - 3 ? 1.0f - fx : big;
        const float ly = cy >= 2 ? fy : big, hy = cy <= g.gy - 3 ? 1.0f - fy : big;
        const float lz = cz >= 2 ? fz : big, hz = cz <= g.gz - 3 ? 1.0f - fz : big;
        const float mf = fmaxf(fminf(fminf(fminf(lx, hx), fminf(ly, hy)), fminf(lz, hz)), 0.0f);
        const bool covers = (cx - 1 <= 0) && (cx + 1 >= g.gx - 1) && (cy - 1 <= 0) && (cy + 1 >= g.gy - 1) &&
                            (cz - 1 <= 0) && (cz + 1 >= g.gz - 1);
        const float bound = (1.0f + mf - 2e-3f) * g.h;
        refine = !(covers || best <= bound * bound + out2 || (max_dist >= 0.0f && bound > max_dist));
    }
}

// per-pair terms -> per-lane f32 accumulators (shared by the main and the refine kernel)
template <bool P2PLANE, int NACC>
__device__ __forceinline__ void accumulate_pair(const GridGeom &g, float (&acc)[NACC], float x, float y, float z,
                                                const float4 &c, const float4 &n) {
    if (P2PLANE) {
        // registration.rs:417-427 in f32: c = s x n ; a = [c, n] ; b = n . (d - s) -- the pair's terms as the reference forms them.
        // Their products go into the lane's sums by FMA (round 5): the reference adds its pairs one after the other in f32, these
        // sums are per lane, then f64 -- another order either way --, so a product that is not rounded on its own is one rounding
        // less, not a departure; 28 instructions per pair instead of 56.
        const float a[6] = {y * n.z - z * n.y, z * n.x - x * n.z, x * n.y - y * n.x, n.x, n.y, n.z};
        const float dx = c.x - x, dy = c.y - y, dz = c.z - z;
        const float b = n.x * dx + n.y * dy + n.z * dz;
        int o = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = r; cc < 6; ++cc) { acc[o] = __builtin_fmaf(a[r], a[cc], acc[o]); ++o; }
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[21 + r] = __builtin_fmaf(a[r], b, acc[21 + r]);
        acc[27] = __builtin_fmaf(b, b, acc[27]);
        acc[28] += 1.0f;
    } else {
        // shifted by the target bbox centre so that H = sum s q^T - n ms mq^T does not cancel
        const float sx = x - g.cx, sy = y - g.cy, sz = z - g.cz;
        const float tx = c.x - g.cx, ty = c.y - g.cy, tz = c.z - g.cz;
        acc[0] += sx; acc[1] += sy; acc[2] += sz;
        acc[3] += tx; acc[4] += ty; acc[5] += tz;
        acc[6] = __builtin_fmaf(sx, tx, acc[6]);   acc[7] = __builtin_fmaf(sx, ty, acc[7]);   acc[8] = __builtin_fmaf(sx, tz, acc[8]);
        acc[9] = __builtin_fmaf(sy, tx, acc[9]);   acc[10] = __builtin_fmaf(sy, ty, acc[10]); acc[11] = __builtin_fmaf(sy, tz, acc[11]);
        acc[12] = __builtin_fmaf(sz, tx, acc[12]); acc[13] = __builtin_fmaf(sz, ty, acc[13]); acc[14] = __builtin_fmaf(sz, tz, acc[14]);
        const float ex = x - c.x, ey = y - c.y, ez = z - c.z;          // registration.rs:214
        acc[15] += ex * ex + ey * ey + ez * ez;
        acc[16] += 1.0f;
    }
}

// ---- GICP pair terms (gicp.rs:205-251), f32 and in the reference's operation order ----------------
// rotation matrix of a unit quaternion (nalgebra UnitQuaternion::to_rotation_matrix)
__device__ __forceinline__ void quat_to_rot(const float q[4], float R[9]) {
    const float i = q[0], j = q[1], k = q[2], w = q[3];
    const float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    const float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    R[0] = ww + ii - jj - kk; R[1] = ij - wk; R[2] = wj + ik;
    R[3] = wk + ij; R[4] = ww - ii + jj - kk; R[5] = jk - wi;
    R[6] = ik - wj; R[7] = wi + jk; R[8] = ww - ii - jj + kk;
}

__device__ __forceinline__ void mat3_mul_f(const float a[9], const float b[9], float c[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) c[3 * r + cc] = (a[3 * r] * b[cc] + a[3 * r + 1] * b[3 + cc]) + a[3 * r + 2] * b[6 + cc];
}

// covariances are stored as two float4 per point: (xx, xy, xz, yy), (yz, zz, -, -)
__device__ __forceinline__ void cov_unpack(const float4 &a, const float4 &b, float c[9]) {
    c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.y; c[4] = a.w; c[5] = b.x; c[6] = a.z; c[7] = b.x; c[8] = b.y;
}

// acc layout = the 29 words the point-to-plane finalize step reads: lower triangle of H by columns
// (entry (row c, col r), r <= c, at the slot of the upper-triangle walk r, c), g, sum dist^2, count
__device__ __forceinline__ void accumulate_gicp(float (&acc)[TC_ICP_SUMS_P2PLANE], float x, float y, float z, const float4 &c, float d2,
                                                const float R[9], const float4 &cs0, const float4 &cs1, const float4 &ct0, const float4 &ct1) {
    float Cs[9], Ct[9], Rt[9], tmp[9], rcr[9], M[9], Mi[9];
    cov_unpack(cs0, cs1, Cs);
    cov_unpack(ct0, ct1, Ct);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) Rt[3 * r + cc] = R[3 * cc + r];
    mat3_mul_f(R, Cs, tmp);
    mat3_mul_f(tmp, Rt, rcr);
#pragma unroll
    for (int e = 0; e < 9; ++e) M[e] = Ct[e] + rcr[e];                            // :217
    // nalgebra Matrix3::try_inverse: adjugate / determinant; a zero determinant skips the pair (:218-221)
    const float m11 = M[0], m12 = M[1], m13 = M[2], m21 = M[3], m22 = M[4], m23 = M[5], m31 = M[6], m32 = M[7], m33 = M[8];
    const float mi_a = m22 * m33 - m32 * m23, mi_b = m21 * m33 - m31 * m23, mi_c = m21 * m32 - m31 * m22;
    const float det = (m11 * mi_a - m12 * mi_b) + m13 * mi_c;
    if (det == 0.0f) return;
    Mi[0] = mi_a / det; Mi[1] = (m13 * m32 - m33 * m12) / det; Mi[2] = (m12 * m23 - m22 * m13) / det;
    Mi[3] = -mi_b / det; Mi[4] = (m11 * m33 - m31 * m13) / det; Mi[5] = (m13 * m21 - m23 * m11) / det;
    Mi[6] = mi_c / det; Mi[7] = (m12 * m31 - m32 * m11) / det; Mi[8] = (m11 * m22 - m21 * m12) / det;
    const float r[3] = {c.x - x, c.y - y, c.z - z};                                // residual t - T s
    const float a[9] = {-0.0f, z, -y, -z, -0.0f, x, y, -x, -0.0f};                 // A = -skew(T s)
    float at[9], mia[9], hrr[9], hrt[9];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) at[3 * rr + cc] = a[3 * cc + rr];
    mat3_mul_f(Mi, a, mia);
    mat3_mul_f(at, mia, hrr);
    mat3_mul_f(at, Mi, hrt);
    float wr[3], gr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wr[i] = (Mi[3 * i] * r[0] + Mi[3 * i + 1] * r[1]) + Mi[3 * i + 2] * r[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) gr[i] = (at[3 * i] * wr[0] + at[3 * i + 1] * wr[1]) + at[3 * i + 2] * wr[2];
    int o = 0;
#pragma unroll
    for (int rr = 0; rr < 6; ++rr)
#pragma unroll
        for (int cc = rr; cc < 6; ++cc) {           // lower-triangle entry (row cc, column rr) of the reference's H
            float v;
            if (cc < 3) v = hrr[3 * cc + rr];
            else if (rr < 3) v = hrt[3 * rr + (cc - 3)];
            else v = Mi[3 * (cc - 3) + (rr - 3)];
            acc[o] += v;
            ++o;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[21 + i] += gr[i]; acc[24 + i] += wr[i]; }
    const float dist = sqrtf(d2);
    acc[27] += dist * dist;                                                        // :250
    acc[28] += 1.0f;
}

// Main pass: ring-1 search (warm-start pruned).  Lanes whose ring-1 answer is not provably exact
// (Poisson tail, queries outside the target's box) are NOT finished here: they go to a list served
// by icp_refine_kernel in dense waves.
//
// The kernel is bound by loads in flight = waves per SIMD = VGPRs, so the work of a group of
// kIcpGroup x 256 points is split in two phases whose register sets do not add up:
//   S  search: one point at a time, result -> corr_pos (no accumulators live);
//   A  accumulate: the matched target records + normals of the lane's points are gathered together
//      (independent loads, one round trip) and folded into the f32 per-lane sums, which are then
//      wave-reduced into the block's f64 LDS row, so that nothing but that row survives the group.
constexpr int kIcpGroup = 4;
constexpr int kSearchersCol = TC_ICP_SUMS_STRIDE - 1;          // spare column of the per-block rows: lanes that searched (icp_correspond_reduce_kernel)
static_assert(kSearchersCol >= TC_ICP_SUMS_P2PLANE && kSearchersCol >= TC_ICP_SUMS_P2P, "the searcher count needs a column the sums do not use");
// Dense search trips (round 6, the certificate's instantiation): the lanes of a wave that have to search -- scattered over the group's
// four trips, each trip a chain of dependent reads that the wave walks for one searching lane as for sixty-four -- are packed through
// LDS into ceil(S / 64) trips of 64 whenever that halves the number of trips.  Same queries, same searches, same results handed back
// in the same (trip, lane) order: same bits.  With the second-neighbour certificate a converged TUM-shaped pair has ~2 searching lanes
// per wave: ONE trip instead of up to four (profiles/r06_dense_trips_certificate.txt).
constexpr int kRefineBlocks = 256;          // blocks of the refine pass = rows handed to the finalize step
// The refine list: one count per wave of a main block (kMaxPartialBlocks x 4 words), then the entries, 32 bytes each:
//   {x, y, z (the transformed query), source index} {best known d2, its position, -, -}
// so that the refine pass starts from ONE 32-byte read instead of entry -> source record + previous match -> transform.
constexpr int kRefineEntryWords = 8;
__host__ __device__ __forceinline__ uint4 *refine_entries(uint32_t *rlist) {
    uintptr_t a = reinterpret_cast<uintptr_t>(rlist + kMaxPartialBlocks * (kIcpBlock / 64));
    return reinterpret_cast<uint4 *>((a + 15) & ~(uintptr_t)15);
}

// MODE: 0 point-to-point, 1 point-to-plane (tgt_nrm = target normals in cell order), 2 GICP (tgt_nrm = target
// covariances, two float4 per cell-sorted position; src_cov = source covariances in the source's sorted order)
// 4 waves per SIMD = the launch geometry (one round of 1024 blocks of 4 waves on 1024 SIMDs): the register allocator may use up
// to 128 VGPRs and must not use more (a 3-wave kernel needs a second round of blocks: +20-40 %).  Measured the other way too:
// pinned to 5 / 6 waves (96 / 80 VGPRs, 96 / 192 bytes of scratch) with 5 / 6 blocks per CU the pass takes 54.6 / 68.4 us
// instead of 46.5: the kernel is register limited, spills cost more than the extra waves hide.
// STATS: the counting instantiation (tc_profile_enable(ctx, 3) / TC_DEBUG & 8): wave trips, trips without a search, searches,
// candidate steps the lanes needed, candidate steps the trips took (their slowest lane) -> IcpState::refine_ring_hist[2..6].
// The product's instantiation carries none of it.
// CERT: the instantiation that maintains and uses the second-neighbour certificate (round 6; chosen by the host per chunk of
// iterations from the word the finalize launch leaves in the pinned block: run_chunked).  The plain instantiation holds none of it.
// COUNT: the instantiation that counts its searching lanes into the rows' spare column (the certificate's gate is evaluated by the finalize
// launch that ends a chunk of iterations, so only a chunk's LAST main pass has to count; the counter costs the plain pass a scalar
// register and 0.3 us -- profiles/r06_dense_trips_certificate.txt -- which the other 43 of 50 launches now do not pay: without it the plain
// kernel is round 5's, instruction for instruction).  The certificate's instantiation always counts.
template <int MODE, bool STATS = false, bool CERT = false, bool COUNT = false>
__global__ void __launch_bounds__(kIcpBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) icp_correspond_reduce_kernel(
    GridView tgt, const float4 *__restrict__ tgt_nrm, float4 *wsrc, uint32_t ns, uint32_t chunk,
    const IcpState *__restrict__ st, uint32_t *__restrict__ rlist,
    double *__restrict__ partials, int dbg, const float4 *__restrict__ src_cov, const float4 *__restrict__ vor,
    unsigned long long *__restrict__ blk_times, const float *__restrict__ pts12) {
    // (the certificate's bound records sit behind the working copy of the source, icp_setup: no kernel parameter of their own -- the plain
    // instantiation keeps round 5's signature)
    [[maybe_unused]] float4 *const wl = CERT ? wsrc + ns + 4 : nullptr;
    constexpr bool P2PLANE = MODE == 1;
    unsigned long long t_begin = 0;
    if (blk_times) t_begin = __builtin_amdgcn_s_memrealtime();          // TC_DEBUG & 1024: per-block start / end stamps (100 MHz)
    constexpr int NACC = MODE == 0 ? TC_ICP_SUMS_P2P : TC_ICP_SUMS_P2PLANE;
    // The block's first group of source records is REQUESTED before the state header is waited for (round 5): the header was
    // written by the previous launch's one-block solve, its read is served from beyond this XCD's L2 like the records' -- two
    // round trips in a row at the head of every block of a 32 us launch.  (A finished registration's launches read 16 KB per block
    // for nothing: they are not on any timed path.)
    const uint32_t lb = xcd_remap_icp(blockIdx.x, gridDim.x);
    const uint32_t beg = lb * chunk;
    const uint32_t end = min(beg + chunk, ns);
    u32x4 first[kIcpGroup];
#pragma unroll
    for (int u = 0; u < kIcpGroup; ++u) {
        const uint32_t j = beg + u * kIcpBlock + threadIdx.x;
        first[u] = __builtin_amdgcn_raw_buffer_load_b128(raw_rsrc(wsrc), (j < end ? j : min(beg, ns - 1)) << 4, 0, 0);
    }
    asm volatile("" ::: "memory");          // (the compiler otherwise sinks the four reads below the header's wait and the early return)
    const IcpHeader hd = load_header(st);
    if (hd.done) return;
    __shared__ double red[kIcpBlock / 64][TC_ICP_SUMS_STRIDE];
    __shared__ uint2 spans[kSpanRows][kIcpBlock];
    // the searching lanes of a wave's group, packed (dense trips below): {x, y, z, budget} in, {best, position, refine flag, bound} back
    // (the certificate's instantiation only: 16 KB per block)
    __shared__ float4 qbuf[CERT ? kIcpBlock / 64 : 1][CERT ? kIcpGroup * 64 : 1];
#ifdef TC_ICP_LDS_PAD
    __shared__ uint32_t lds_pad[TC_ICP_LDS_PAD / 4];          // occupancy probe: fewer blocks fit a CU
    if (hd.iterations == 0xFFFFFFFFu) lds_pad[threadIdx.x] = threadIdx.x;
#endif
    // Refine entries (source index, best known position) go straight to global memory: every wave owns a region
    // of chunk / 4 entries (it never handles more points than that) and appends in (trip, lane) order by ballot
    // prefix, so the list -- and with it every sum -- is deterministic, without atomics, LDS or barriers.
    constexpr int kWavesPerBlock = kIcpBlock / 64;
    const uint32_t wave_cap = chunk / kWavesPerBlock;
    uint4 *__restrict__ const wseg = refine_entries(rlist) + 2 * ((size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * wave_cap;
    uint32_t wcnt = 0;                                     // this wave's entries so far (wave-uniform)
    [[maybe_unused]] uint32_t wsearch = 0;                 // lanes of this wave that had to search (wave-uniform): the rows' spare column (COUNT / CERT)
    const GridGeom &g = tgt.g;
    const float q[4] = {hd.q[0], hd.q[1], hd.q[2], hd.q[3]};
    const float t[3] = {hd.t[0], hd.t[1], hd.t[2]};
    const float max_dist = hd.max_dist;
    const bool warm = hd.iterations > 0 && !(dbg & 1);
    // The second-neighbour certificate (round 5).  wl[j] = (x_ref, L): where source point j stood when it last searched, and a lower
    // bound L of the distance from THERE to every target point other than the match it found.  Wherever the point stands now,
    // every other target point is at least L - |x - x_ref| away: a match closer than that is THE nearest neighbour, strictly -- no
    // search.  (The displacement is measured, not accumulated from per-update bounds: a converged transform still wobbles by ~1e-5
    // per update, and a bound eroded by that every pass sends a few per cent of the lanes back to search in every pass.)  The inscribed-ball test needs |T s - p| < d_nn(p) / 2, which sensor noise of the order of the
    // point spacing defeats for good (TUM-shaped pair: 85 % of the points searched in every iteration, converged or not); this one
    // needs |T s - p| < (distance to the runner-up) - delta, which holds for every point that is not an exact tie once the
    // transform has stopped.  Off (d_ang < 0) while the update is large; TC_DEBUG & 4096 switches it off altogether (A/B).
    const bool track = CERT && TC_ICP_STEP == 4 && warm && wl != nullptr && hd.d_ang >= 0.0f && hd.d_run >= 1u && !(dbg & 4096);
    // (the bounds in wl are the previous pass's only if that pass maintained them too: the switch may flip on and off while the
    // registration hovers around the threshold, and a bound that slept through a large update is no bound)
    const bool use_bounds = track && hd.d_run >= 2u;
    // what the roundings of two passes' query positions can differ by, on top of the update itself: a few ulps of |x| + |t| each
    const float t_norm = sqrtf(hd.t[0] * hd.t[0] + hd.t[1] * hd.t[1] + hd.t[2] * hd.t[2]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < TC_ICP_SUMS_STRIDE) red[w][lane] = 0.0;          // each wave owns one row

    // One contiguous chunk per block.  Measured alternatives (TC_DEBUG & 1024 stamps every block): with one round of 984 blocks
    // the slowest block takes 1.17x (clouds moving) to 1.6x (aligned) the median, all of them start within 0.4 us.  That spread is
    // NOT the work: spreading every block's four 256-point sub-chunks over its XCD's whole slab left it unchanged (p10 / p50 / max
    // 39.8 / 42.9 / 49.2 us), the eight XCDs finish within 2 us of each other; it is how blocks share a CU.  Smaller blocks in
    // several rounds lose more than the tail gives back (512 points per block: 48.7 us, 256: 55.6 us vs 46.9 us).
#ifdef TC_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t gb = beg; gb < end; gb += kIcpGroup * kIcpBlock) {
        // A group = kIcpGroup points per lane, in four stages.  The converged phase of a registration is LATENCY bound (a kept
        // match needs ~150 instructions and three dependent memory round trips), so the loads of the group's points go out
        // together: stage 1 the source records + previous matches, stage 2 the matched target records + their inscribed-ball
        // bounds, stage 4 the final records + normals; only stage 3, the search of the lanes that need one, runs point by point.
        float px[kIcpGroup], py[kIcpGroup], pz[kIcpGroup], ubp[kIcpGroup];
        uint32_t pjv[kIcpGroup], mv[kIcpGroup];
        bool fin[kIcpGroup];
        float4 pv[kIcpGroup];                 // the previous matches' records: reused in stage 4 when the match did not change
        {   // ---- stage 1 + 2 ----
            float4 sv[kIcpGroup];
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                const uint32_t j = gb + u * kIcpBlock + threadIdx.x;
                const bool in = j < end;
                // ONE 16-byte read per point: the source record and, in its w, the position of the previous match (the loop's working
                // copy of the source, icp_working_source_kernel: round 5 -- a read instruction less per point, 4 of ~23; the pass is
                // bound by look-ups per lane and read instruction, profiles/r05_ab_pack12.txt)
                { const u32x4 t4 = first[u];
                  sv[u] = make_float4(__uint_as_float(t4.x), __uint_as_float(t4.y), __uint_as_float(t4.z), __uint_as_float(t4.w)); }
                uint32_t pj = __float_as_uint(sv[u].w);        // (0xFFFFFFFF before iteration 1)
                if (!warm || !in) pj = 0xFFFFFFFFu;
                pjv[u] = pj;
                fin[u] = in && sv[u].x < 3.0e38f;              // a non-finite source point (placeholder record, grid.hip) has no match
            }
            // the previous match's record: from `vor` when it exists -- the same x, y, z with the inscribed-ball bound in w, ONE
            // 16-byte gather instead of record + bound from two arrays (the pass is bound by cache-line accesses as much as by
            // instructions) -- else from the index (its w is the original index: no bound yet)
            const float4 *__restrict__ prec = vor != nullptr ? vor : tgt.pts;
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) pv[u] = prec[pjv[u] != 0xFFFFFFFFu ? pjv[u] : 0u];
            float4 lw[kIcpGroup];              // (track) the points' reference positions and bounds: one coalesced 16-byte read each
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                lw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (use_bounds) { const uint32_t j = gb + u * kIcpBlock + threadIdx.x; lw[u] = wl[j < end ? j : beg]; }
            }
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                iso_apply(q, t, sv[u].x, sv[u].y, sv[u].z, px[u], py[u], pz[u]);
                // warm start: the previous match is a real target point, its distance bounds the new nearest-neighbour distance
                const float d = pjv[u] != 0xFFFFFFFFu ? d2_nc(pv[u].x, pv[u].y, pv[u].z, px[u], py[u], pz[u]) : INFINITY;
                const float vr = (vor != nullptr && pjv[u] != 0xFFFFFFFFu) ? pv[u].w : 0.0f;
                // Inscribed-ball test (exact): vor[p] = a quarter of the squared distance from target point p to its nearest OTHER
                // target point (shaved).  |T s - p| < d_nn(p) / 2 puts every other target point t at |T s - t| >= d_nn(p) - |T s - p|
                // > |T s - p|: the previous match is still THE nearest neighbour, nothing has to be searched.  The flag travels
                // in the sign bit of the (non-negative) bound.
                bool kept = d < vr;
                if (track) {
                    // how far the point stands from where its bound was measured, plus what the roundings of the two positions can
                    // differ by (a few ulps of |x| + |t| each); the comparison is made on squares with a relative margin of 2e-5
                    const float ex = px[u] - lw[u].x, ey = py[u] - lw[u].y, ez = pz[u] - lw[u].z;
                    const float r = __builtin_amdgcn_sqrtf(px[u] * px[u] + py[u] * py[u] + pz[u] * pz[u]);
                    const float moved = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.00001f + 3e-6f * (r + t_norm);
                    const float lm = lw[u].w - moved;
                    const bool cert = use_bounds && lm > 0.0f && d * 1.00002f < lm * lm;
                    // the first pass after the switch came on: whatever the record held is void (matches may have changed while nobody
                    // kept it); a lane that goes on to search writes a fresh one behind its search
                    const uint32_t j = gb + u * kIcpBlock + threadIdx.x;
                    if (!use_bounds && j < end) wl[j] = make_float4(px[u], py[u], pz[u], 0.0f);
                    kept = kept || cert;
                }
                ubp[u] = kept ? -d : d;
            }
        }
        // ---- stage 3 ----
#ifdef TC_PHASE_STAMPS
        { float sink = px[0] + py[1] + pz[2] + ubp[3] + ubp[0] + ubp[1] + ubp[2]; asm volatile("" :: "v"(sink)); }
#endif
        TC_STAMP(0);
        // ---- stage 3: the searches of the group.  Which lanes search is known for all four trips at once; when packing them halves
        // the number of trips (dense form, below) the wave searches in dense trips through LDS, else trip by trip in place ----
        // (the plain instantiation computes none of this: it ballots trip by trip, as before)
        [[maybe_unused]] unsigned long long smv[kIcpGroup] = {0ull, 0ull, 0ull, 0ull};
        [[maybe_unused]] uint32_t sbase[kIcpGroup + 1] = {0u, 0u, 0u, 0u, 0u};
        [[maybe_unused]] uint32_t nsearch = 0u;
        [[maybe_unused]] bool dense = false;
        [[maybe_unused]] const unsigned long long lt_mask = (1ull << lane) - 1ull;
        if constexpr (CERT) {
            uint32_t trips_in_place = 0u;
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                const bool keep = __float_as_uint(ubp[u]) >> 31;
                smv[u] = __ballot(fin[u] && !keep);
                sbase[u + 1] = sbase[u] + (uint32_t)__popcll(smv[u]);
                trips_in_place += smv[u] != 0ull ? 1u : 0u;
            }
            nsearch = sbase[kIcpGroup];
            dense = 2u * ((nsearch + 63u) / 64u) <= trips_in_place;
        }
        if constexpr (CERT) if (dense) {
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                const bool keep = __float_as_uint(ubp[u]) >> 31;
                float ub2 = fabsf(ubp[u]);
                if (track) ub2 *= 6.25f;          // (a search that is to MEASURE the second-neighbour bound looks 2.5 times as far: see the in-place form)
                if (max_dist >= 0.0f) ub2 = fminf(ub2, max_dist * max_dist * 1.0001f);
                if (fin[u] && !keep) qbuf[w][sbase[u] + (uint32_t)__popcll(smv[u] & lt_mask)] = make_float4(px[u], py[u], pz[u], ub2);
            }
            // (a wave's own LDS traffic is served in order: its lanes read what other lanes of the same wave wrote, no block barrier)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
            for (uint32_t tb = 0; tb < nsearch; tb += 64u) {
                const uint32_t qi = tb + (uint32_t)lane;
                const bool act = qi < nsearch;
                const float4 qv = qbuf[w][act ? qi : tb];
                float best = INFINITY;
                uint32_t bestg = 0xFFFFFFFFu;
                bool refine = false;
                uint32_t nst = 0u;
                float low2 = 0.0f;
                nn_search_pruned<STATS, CERT>(tgt, qv.x, qv.y, qv.z, act ? qv.w : -1.0f, best, bestg, refine, max_dist, spans, nst, pts12, track, low2 TC_STAMP_PASS);
                if (act) qbuf[w][qi] = make_float4(best, __uint_as_float(bestg), __uint_as_float(refine ? 1u : 0u), low2);
                if constexpr (STATS) {
                    uint32_t mx = act ? nst : 0u, sm = act ? nst : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); sm += (uint32_t)__shfl_xor((int)sm, o); }
                    if (lane == 0) {
                        IcpState *sw = const_cast<IcpState *>(st);
                        atomicAdd(&sw->refine_ring_hist[2], mx);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&sw->refine_ring_hist[3]), (unsigned long long)sm);
                        atomicAdd(&sw->refine_ring_hist[7], min(nsearch - tb, 64u));
                        atomicAdd(&sw->refine_ring_hist[5], 1u);
                    }
                }
            }
            if constexpr (STATS) {          // (the group's trips that did not have to be taken count as trips without a search)
                const uint32_t taken = (nsearch + 63u) / 64u;
                if (lane == 0 && taken < (uint32_t)kIcpGroup) {
                    IcpState *sw = const_cast<IcpState *>(st);
                    atomicAdd(&sw->refine_ring_hist[5], (uint32_t)kIcpGroup - taken);
                    atomicAdd(&sw->refine_ring_hist[6], (uint32_t)kIcpGroup - taken);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int u = 0; u < kIcpGroup; ++u) {
            // Wave priority by progress inside the group (round 4; s_setprio takes an immediate: u is a constant after unrolling).
            // The launch is one round of blocks, four waves per SIMD, and under the default oldest-first arbitration they retire one
            // after the other -- block durations p10 / p50 / max 33 / 37.5 / 46 us (TC_DEBUG=1024) -- the last ones at an occupancy
            // that hides no latency.  A wave on its first point issues ahead of one on its fourth: 39.6 -> 38.3 us per pass, ICP
            // alone +2.3 %, the 10 M-point sharded loop (ten groups per block) +5 % (profiles/r04_ab_wave_priority.txt; the
            // opposite order, other level tables, priority by the share of the whole block done, a boost for the window reads:
            // equal or worse).
            if (u == 0) __builtin_amdgcn_s_setprio(3); else if (u == 1) __builtin_amdgcn_s_setprio(2); else if (u == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            const uint32_t j = gb + u * kIcpBlock + threadIdx.x;
            const bool in = fin[u];
            const float x = px[u], y = py[u], z = pz[u];
            const uint32_t pj = pjv[u];
            const bool keep = __float_as_uint(ubp[u]) >> 31;             // (-0.0 too: an exact hit on a non-duplicate point)
            const float ub2p = fabsf(ubp[u]);
            float ub2 = ub2p;
            if (max_dist >= 0.0f) ub2 = fminf(ub2, max_dist * max_dist * 1.0001f);   // farther matches are rejected anyway
            float best = INFINITY;
            uint32_t bestg = 0xFFFFFFFFu;
            bool refine = false;
            uint32_t nst = 0u;
            // (track) a search that is to MEASURE the bound looks 2.5 times as far as the previous match: what it does not scan could
            // be anywhere beyond its budget, and a budget of exactly the match's distance would cap the bound at that distance --
            // the certificate could never fire.  The superset it scans holds the same nearest neighbour.
            if (track) { ub2 = ub2p * 6.25f; if (max_dist >= 0.0f) ub2 = fminf(ub2, max_dist * max_dist * 1.0001f); }
            const unsigned long long smask = CERT ? smv[u] : __ballot(in && !keep);
            if constexpr (COUNT || CERT) wsearch += (uint32_t)__popcll(smask);
            float low2 = 0.0f;
            if (CERT && dense) {
                if (in && !keep) {          // the dense trips' result of this lane's query
                    const float4 rv = qbuf[CERT ? w : 0][CERT ? sbase[u] + (uint32_t)__popcll(smask & lt_mask) : 0u];
                    best = rv.x; bestg = __float_as_uint(rv.y); refine = __float_as_uint(rv.z) != 0u; low2 = rv.w;
                }
            } else {
                if (smask != 0ull) nn_search_pruned<STATS, CERT>(tgt, x, y, z, (keep || !in) ? -1.0f : ub2, best, bestg, refine, max_dist, spans, nst, pts12, track, low2 TC_STAMP_PASS);
                if constexpr (STATS) {
                    uint32_t mx = nst, sm = (in && !keep) ? nst : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); sm += (uint32_t)__shfl_xor((int)sm, o); }
                    if (lane == 0) {
                        IcpState *sw = const_cast<IcpState *>(st);
                        atomicAdd(&sw->refine_ring_hist[2], mx);                               // steps the trip took (its slowest lane)
                        // steps its lanes needed: the one counter that can pass 2^32 inside a call (10 M points x 50 iterations x ~5
                        // steps: ADVICE r5) -> a 64-bit add on the aligned pair [3..4]; the searches moved to [7]
                        atomicAdd(reinterpret_cast<unsigned long long *>(&sw->refine_ring_hist[3]), (unsigned long long)sm);
                        atomicAdd(&sw->refine_ring_hist[7], (uint32_t)__popcll(smask));        // searches (<= points x iterations / call: checked on the host)
                        atomicAdd(&sw->refine_ring_hist[5], 1u);                               // wave trips
                        if (smask == 0ull) atomicAdd(&sw->refine_ring_hist[6], 1u);            // ... without a search
                    }
                }
            }
            // (track) a lane that searched has measured its bound afresh: the distance, a hair short (v_sqrt_f32 is good to an ulp)
            if (track && in && !keep && j < end) wl[j] = make_float4(x, y, z, refine ? 0.0f : __builtin_amdgcn_sqrtf(fmaxf(low2, 0.0f)) * 0.999999f);
            if (keep) { best = ub2p; bestg = pj; }
            refine = refine && in && !keep;
            const unsigned long long rmask = __ballot(refine);
            if (refine) {
                // hand the refine pass the best real point seen so far: previous match or ring-1 best
                // (everything of ring 1 inside the ball has been examined: the refine pass starts at ring 2)
                const unsigned long long kp = pj != 0xFFFFFFFFu ? (((unsigned long long)__float_as_uint(ub2p) << 32) | pj) : ~0ull;
                const unsigned long long kb = bestg != 0xFFFFFFFFu ? (((unsigned long long)__float_as_uint(best) << 32) | bestg) : ~0ull;
                const unsigned long long km = kb < kp ? kb : kp;
                uint4 *ent = wseg + 2 * (size_t)(wcnt + __popcll(rmask & ((1ull << lane) - 1ull)));
                ent[0] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), j);
                ent[1] = make_uint4((uint32_t)(km >> 32), km != ~0ull ? (uint32_t)km : 0xFFFFFFFFu, 0u, 0u);
            }
            wcnt += __popcll(rmask);
            bool valid = in && !refine && bestg != 0xFFFFFFFFu;
            // registration.rs:100-101.  A real (scalar) branch on the wave-uniform cut-off: if-converted, the correctly rounded sqrt
            // -- a dozen instructions per point -- ran in every call without a cut-off (the empty asm keeps it from being speculated)
            if (max_dist >= 0.0f) { asm volatile(""); if (valid) valid = !(sqrtf(best) > max_dist); }
            const uint32_t newv = valid ? bestg : 0xFFFFFFFFu;
            if (j < end && !refine && (!warm || newv != pj)) reinterpret_cast<uint32_t *>(wsrc)[4 * (size_t)j + 3] = newv;     // an unchanged match is not written again
            mv[u] = (dbg & 16) ? 0xFFFFFFFFu : newv;                             // dbg & 16: timing experiments only (no sums)
            TC_STAMP(3);
        }
        // ---- stage 4 ----: per-pair terms of the group -> per-lane f32 sums -> the wave's f64 row
        float acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
        {
            float4 cv[kIcpGroup], nv[kIcpGroup];
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                const uint32_t m = mv[u] != 0xFFFFFFFFu ? mv[u] : 0u;
                cv[u] = pv[u];
                // (a changed match's record comes from the PACKED copy the candidate loop has just read it from -- a line that is in L1 --
                // not from the 16-byte records, which the pass then does not touch at all: one array less in the caches)
                if (mv[u] != pjv[u]) { const f32x3 t3 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(raw_rsrc(pts12), (m << 3) + (m << 2), 0, 0));
                                       cv[u] = make_float4(t3.x, t3.y, t3.z, 0.0f); }
                nv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == 1) { const f32x3 t3 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(raw_rsrc(tgt_nrm), m << 4, 0, 0));
                                 nv[u] = make_float4(t3.x, t3.y, t3.z, 0.0f); }
            }
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                if (mv[u] == 0xFFFFFFFFu) continue;
                if constexpr (MODE == 2) {
                    const uint32_t j = gb + u * kIcpBlock + threadIdx.x;
                    float R[9];
                    quat_to_rot(q, R);
                    accumulate_gicp(acc, px[u], py[u], pz[u], cv[u], d2_nc(cv[u].x, cv[u].y, cv[u].z, px[u], py[u], pz[u]), R, src_cov[2 * (size_t)j],
                                    src_cov[2 * (size_t)j + 1], tgt_nrm[2 * (size_t)mv[u]], tgt_nrm[2 * (size_t)mv[u] + 1]);
                } else {
                    accumulate_pair<P2PLANE, NACC>(g, acc, px[u], py[u], pz[u], cv[u], nv[u]);
                }
            }
        }
#ifdef TC_PHASE_STAMPS
        { float sink = 0.f; for (int i = 0; i < NACC; ++i) sink += acc[i]; asm volatile("" :: "v"(sink)); }
#endif
        TC_STAMP(4);
        // per-group fold: transposing wave reduction (f32, fixed tree) -> this wave's f64 row
        wave_fold_transposed<NACC>(acc, red[w], lane);
        TC_STAMP(5);
        // the next group's source records (blocks of more than one group: the 10 M-point loop), requested where nothing else is live
        if (gb + kIcpGroup * kIcpBlock < end) {
#pragma unroll
            for (int u = 0; u < kIcpGroup; ++u) {
                const uint32_t j = gb + (kIcpGroup + u) * kIcpBlock + threadIdx.x;
                first[u] = __builtin_amdgcn_raw_buffer_load_b128(raw_rsrc(wsrc), (j < end ? j : beg) << 4, 0, 0);
            }
        }
    }
    // column kSearchersCol of the row (beyond the sums' columns, which end at 29): how many lanes of the block searched -- summed with
    // the rest by the refine fold and icp_finalize, which leaves the total in IcpState::searchers (the certificate's gate, compose())
    if constexpr (COUNT || CERT) { if (lane == kSearchersCol) red[w][kSearchersCol] = (double)wsearch; }
    __syncthreads();
    if (threadIdx.x < TC_ICP_SUMS_STRIDE) {
        double sum = 0.0;
        if (threadIdx.x < NACC || ((COUNT || CERT) && threadIdx.x == kSearchersCol)) {
#pragma unroll
            for (int w2 = 0; w2 < kIcpBlock / 64; ++w2) sum += red[w2][threadIdx.x];
        }
        partials[(size_t)blockIdx.x * TC_ICP_SUMS_STRIDE + threadIdx.x] = sum;
    }
    if (lane == 0) rlist[blockIdx.x * kWavesPerBlock + w] = wcnt;
    if (blk_times && threadIdx.x == 0) {
        blk_times[2 * blockIdx.x] = t_begin;
        blk_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#ifdef TC_PHASE_STAMPS
        TC_STAMP(6);
        for (int i = 0; i < 8; ++i) blk_times[2 * kMaxPartialBlocks + 8 * blockIdx.x + i] = ph[i];
#endif
    }
}

// Refine pass: the listed queries (a few thousand per iteration: Poisson tail, queries outside the
// target's box) continue with ring 2, 3, ... , kRG lanes per query.  A shell is visited as the cells
// the ball |p - q| <= best can reach (ball pruning; a shell that misses the ball ends the search).
// The pass is latency bound (few waves, every query a chain of dependent reads), so the chain is kept
// short: the list entry carries (source index, best known target position), a lane takes four shell
// cells at a time with all their cell_start reads, then all their record reads, in flight together,
// and the shell enumeration of ring 2 is compile-time.  Per-block sums go to the partial rows after
// the main pass's rows.
constexpr int kRefineThreads = 1024;
#ifndef TC_REFINE_LANES
#define TC_REFINE_LANES 32
#endif
constexpr int kRG = TC_REFINE_LANES;                     // lanes per query (measured on the benchmark: 16 lanes 15.0 us, 64 lanes 13.0 us, 32: 11.8 us)
constexpr int kRB = 4;                      // shell cells per lane and batch

__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = kRG / 2; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}

// offset of cell `idx` of the shell of radius R (W = 2R+1, V = W-2): two z faces (W x W), two y
// faces (W x V), two x faces (V x V)
__device__ __forceinline__ void shell_cell(int idx, int R, int W, int V, int &ox, int &oy, int &oz) {
    const int nz = 2 * W * W, ny = 2 * W * V;
    if (idx < nz) {
        const int f = idx / (W * W), r = idx - f * W * W;
        oz = f ? R : -R; oy = r / W - R; ox = r - (r / W) * W - R;
    } else if (idx < nz + ny) {
        const int i2 = idx - nz, f = i2 / (W * V), r = i2 - f * W * V;
        oy = f ? R : -R; oz = r / W - (R - 1); ox = r - (r / W) * W - R;
    } else {
        const int i3 = idx - nz - ny, f = i3 / (V * V), r = i3 - f * V * V;
        ox = f ? R : -R; oz = r / V - (R - 1); oy = r - (r / V) * V - (R - 1);
    }
}

// one shell: returns the group's best key of the shell; touched = some cell of the shell lies in the ball
template <int RC>      // RC > 0: compile-time radius (divisions by constants); RC == 0: run-time R
__device__ __forceinline__ unsigned long long refine_shell(const GridView &tgt, const __amdgpu_buffer_rsrc_t &cs_rsrc,
                                                           const __amdgpu_buffer_rsrc_t &pt_rsrc, int Rrt, int lg, float x, float y,
                                                           float z, int cx, int cy, int cz, unsigned long long bestkey, bool &touched) {
    const GridGeom &g = tgt.g;
    const int R = RC > 0 ? RC : Rrt;
    const int W = 2 * R + 1, V = W - 2;
    const int ncell = 2 * W * W + 2 * W * V + 2 * V * V;
    const float bestd = __uint_as_float((uint32_t)(bestkey >> 32));     // NaN pattern when empty
    const bool have = bestkey != ~0ull;
    unsigned long long lk = ~0ull;
    touched = false;
    for (int idx0 = lg; idx0 < ncell; idx0 += kRG * kRB) {
        uint32_t s0[kRB], e0[kRB];
#pragma unroll
        for (int t = 0; t < kRB; ++t) {
            const int idx = idx0 + t * kRG;
            int ox = 0, oy = 0, oz = 0;
            shell_cell(min(idx, ncell - 1), R, W, V, ox, oy, oz);
            const int ccx = cx + ox, ccy = cy + oy, ccz = cz + oz;
            bool in = idx < ncell && ccx >= 0 && ccx < g.gx && ccy >= 0 && ccy < g.gy && ccz >= 0 && ccz < g.gz;
            if (in && have) {
                const float gx = axis_gap(x, g.minx, g.h, ccx, g.gx - 1, g.clamped), gy = axis_gap(y, g.miny, g.h, ccy, g.gy - 1, g.clamped),
                            gz = axis_gap(z, g.minz, g.h, ccz, g.gz - 1, g.clamped);
                in = !(gx * gx + gy * gy + gz * gz > bestd);               // inside the ball
            }
            touched |= in;
            const uint32_t c = in ? ((uint32_t)ccz * g.gy + ccy) * g.gx + ccx : 0u;
            const uint2 w = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(cs_rsrc, c << 2, 0, 0));
            s0[t] = w.x;
            e0[t] = in ? w.y : w.x;
        }
        // first four records of every cell (reads past a cell hit real points of the next cells or the
        // +inf padding: harmless extra candidates), then the rare long cells
#pragma unroll
        for (int t = 0; t < kRB; ++t) {
            const uint32_t jj = s0[t], o = jj << 4;
            const f32x3 p0 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o, 0, 0));
            const f32x3 p1 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 16u, 0, 0));
            const f32x3 p2 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 32u, 0, 0));
            const f32x3 p3 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 48u, 0, 0));
            const unsigned long long k0 = ((unsigned long long)__float_as_uint(d2_nc(p0.x, p0.y, p0.z, x, y, z)) << 32) | jj;
            const unsigned long long k1 = ((unsigned long long)__float_as_uint(d2_nc(p1.x, p1.y, p1.z, x, y, z)) << 32) | (jj + 1);
            const unsigned long long k2 = ((unsigned long long)__float_as_uint(d2_nc(p2.x, p2.y, p2.z, x, y, z)) << 32) | (jj + 2);
            const unsigned long long k3 = ((unsigned long long)__float_as_uint(d2_nc(p3.x, p3.y, p3.z, x, y, z)) << 32) | (jj + 3);
            const unsigned long long ka = k0 < k1 ? k0 : k1, kb = k2 < k3 ? k2 : k3;
            const unsigned long long kc = ka < kb ? ka : kb;
            if (s0[t] < e0[t]) lk = kc < lk ? kc : lk;
        }
#pragma unroll
        for (int t = 0; t < kRB; ++t) {
            for (uint32_t jj = s0[t] + 4; jj < e0[t]; ++jj) {
                const f32x3 p = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, jj << 4, 0, 0));
                const unsigned long long key = ((unsigned long long)__float_as_uint(d2_nc(p.x, p.y, p.z, x, y, z)) << 32) | jj;
                lk = key < lk ? key : lk;
            }
        }
    }
    return group_min_u64(lk);
}

// Ball scan (round 3): once a query holds a candidate, everything that can beat it lies in the ball of that distance -- and the ball
// clipped to the grid is enumerated DIRECTLY, row by row ((y, z) pairs dealt to the group's lanes, the x window of a row in closed
// form like the normals' pruned scans) instead of shell by shell around the query's cell.  Shells cost O(R^3) cell tests for a
// match R cells away; a query D cells OUTSIDE the target's box has its match in a thin cap of the box (depth ~1 cell, lateral
// radius sqrt(2 D) cells): the rows of the cap, O(D), not the (2 sqrt(2 D))^3 cells of the shells that reach it.  A lane's limit is
// its own best so far, refreshed from the group every few batches: always >= the final distance, so no cell of the final ball is
// skipped.  Exact like the shells (same distance expression, same (distance, position) key).
__device__ __forceinline__ unsigned long long refine_ball_scan(const GridView &tgt, const __amdgpu_buffer_rsrc_t &cs_rsrc,
                                                               const __amdgpu_buffer_rsrc_t &pt_rsrc, int lg, float x, float y, float z,
                                                               unsigned long long bestkey, float cap2) {
    const GridGeom &g = tgt.g;
    float lim = fminf(__uint_as_float((uint32_t)(bestkey >> 32)), cap2);
    // rows the ball can reach (clamped coordinates: with a clamped box the boundary rows also hold the points beyond it).  A query
    // OUTSIDE the box along some axis is at least that far from every record (all of them lie inside an exact box), which leaves
    // the other axes only the rest of the budget: the cap of the box the ball cuts off, not the ball's whole bounding square.
    const float ox = g.clamped ? 0.0f : fmaxf(fmaxf(g.minx - x, x - g.maxx), 0.0f) * 0.9999f;
    const float oy = g.clamped ? 0.0f : fmaxf(fmaxf(g.miny - y, y - g.maxy), 0.0f) * 0.9999f;
    const float oz = g.clamped ? 0.0f : fmaxf(fmaxf(g.minz - z, z - g.maxz), 0.0f) * 0.9999f;
    const float ry = sqrtf(fmaxf(lim - ox * ox - oz * oz, 0.0f)) * 1.0001f + 4e-3f * g.h;
    const float rz = sqrtf(fmaxf(lim - ox * ox - oy * oy, 0.0f)) * 1.0001f + 4e-3f * g.h;
    const int y0 = cell_coord(fminf(fmaxf(y - ry, g.miny), g.maxy), g.miny, g.inv_h, g.gy), y1 = cell_coord(fminf(fmaxf(y + ry, g.miny), g.maxy), g.miny, g.inv_h, g.gy);
    const int z0 = cell_coord(fminf(fmaxf(z - rz, g.minz), g.maxz), g.minz, g.inv_h, g.gz), z1 = cell_coord(fminf(fmaxf(z + rz, g.minz), g.maxz), g.minz, g.inv_h, g.gz);
    const int ny = y1 - y0 + 1;
    const uint32_t nrows = (uint32_t)ny * (uint32_t)(z1 - z0 + 1);
    unsigned long long lk = bestkey;
    uint32_t batch = 0;
    for (uint32_t r0i = (uint32_t)lg; r0i < nrows; r0i += kRG * kRB, ++batch) {
        uint32_t s0[kRB], e0[kRB];
#pragma unroll
        for (int t = 0; t < kRB; ++t) {
            const uint32_t ri = r0i + (uint32_t)t * kRG;
            const int zz = z0 + (int)(min(ri, nrows - 1) / (uint32_t)ny), yy = y0 + (int)(min(ri, nrows - 1) % (uint32_t)ny);
            const float gy = axis_gap(y, g.miny, g.h, yy, g.gy - 1, g.clamped), gz = axis_gap(z, g.minz, g.h, zz, g.gz - 1, g.clamped);
            const float rg = gy * gy + gz * gz;
            bool in = ri < nrows && !(rg + ox * ox > lim);
            int xa = 0, xb = 0;
            if (in) {
                const float rx = __builtin_amdgcn_sqrtf(fmaxf(lim - rg, 0.0f)) * 1.0001f + 4e-3f * g.h;
                const float fa = fminf(fmaxf((x - rx - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
                const float fb = fmaxf(fminf((x + rx - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
                xa = (int)fa; xb = (int)fb;
                in = xa <= xb;
            }
            const uint32_t row = ((uint32_t)zz * g.gy + yy) * g.gx;
            uint32_t s = 0, e = 0;
            if (in) {
                s = __builtin_amdgcn_raw_buffer_load_b32(cs_rsrc, (row + (uint32_t)xa) << 2, 0, 0);
                e = __builtin_amdgcn_raw_buffer_load_b32(cs_rsrc, (row + (uint32_t)xb + 1u) << 2, 0, 0);
            }
            s0[t] = s; e0[t] = e;
        }
#pragma unroll
        for (int t = 0; t < kRB; ++t) {
            for (uint32_t jj = s0[t]; jj < e0[t]; jj += 4) {
                const uint32_t o = jj << 4;
                const f32x3 p0 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o, 0, 0));
                const f32x3 p1 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 16u, 0, 0));
                const f32x3 p2 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 32u, 0, 0));
                const f32x3 p3 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 48u, 0, 0));
                // (reads past the span hit real points of the next cells or the +inf padding: harmless extra candidates)
                const unsigned long long k0 = ((unsigned long long)__float_as_uint(d2_nc(p0.x, p0.y, p0.z, x, y, z)) << 32) | jj;
                const unsigned long long k1 = ((unsigned long long)__float_as_uint(d2_nc(p1.x, p1.y, p1.z, x, y, z)) << 32) | (jj + 1);
                const unsigned long long k2 = ((unsigned long long)__float_as_uint(d2_nc(p2.x, p2.y, p2.z, x, y, z)) << 32) | (jj + 2);
                const unsigned long long k3 = ((unsigned long long)__float_as_uint(d2_nc(p3.x, p3.y, p3.z, x, y, z)) << 32) | (jj + 3);
                const unsigned long long ka = k0 < k1 ? k0 : k1, kb = k2 < k3 ? k2 : k3;
                const unsigned long long kc = ka < kb ? ka : kb;
                lk = kc < lk ? kc : lk;
            }
        }
        lim = fminf(lim, __uint_as_float((uint32_t)(lk >> 32)));
        if ((batch & 3u) == 3u) {       // what the other lanes of the group have found tightens this lane's limit
            float gl = lim;
#pragma unroll
            for (int o = kRG / 2; o > 0; o >>= 1) gl = fminf(gl, __shfl_xor(gl, o));
            lim = gl;
        }
    }
    return group_min_u64(lk);
}

template <int MODE>
__global__ void __launch_bounds__(kRefineThreads) icp_refine_kernel(
    GridView tgt, const float4 *__restrict__ tgt_nrm, const float4 *__restrict__ src,
    IcpState *__restrict__ st, uint32_t *__restrict__ corr_pos /* stride 4: the w of the working source records */, uint32_t *__restrict__ rlist,
    const double *__restrict__ main_rows, uint32_t n_main_rows, uint32_t seg_stride, double *__restrict__ partial_rows,
    const float4 *__restrict__ src_cov, int dbg) {
    constexpr bool P2PLANE = MODE == 1;
    constexpr int NACC = MODE == 0 ? TC_ICP_SUMS_P2P : TC_ICP_SUMS_P2PLANE;
    // Everything the block needs before its first query is REQUESTED before anything is waited for: the state header, its
    // share of the main pass's rows and its four segment counts.  (Until round 4 these were a `done` test, then a loop of
    // four dependent row loads, then two loops of four dependent count loads: ~10 round trips to memory in a row = most of the
    // 5.2 us this launch took with an empty list.)
    const IcpHeader hd = load_header(st);
    const bool dbg_shells = (dbg & 8192) != 0;          // TC_DEBUG & 8192: shells only, as before round 3 (A/B)
    __shared__ float lacc[kRefineThreads / kRG][TC_ICP_SUMS_STRIDE];
    // every refine block also folds its share of the main pass's per-block rows (written by the
    // previous launch) into its own row, in a fixed order: icp_finalize then reads kRefineBlocks rows
    const uint32_t rows_per = (n_main_rows + gridDim.x - 1) / gridDim.x;
    const uint32_t mr0 = min(blockIdx.x * rows_per, n_main_rows), mr1 = min(mr0 + rows_per, n_main_rows);
    constexpr int kFoldBatch = kMaxPartialBlocks / kRefineBlocks;          // rows per block at the largest main grid
    double fr[kFoldBatch];
#pragma unroll
    for (int k = 0; k < kFoldBatch; ++k) {          // (unconditional, clamped reads: a predicated read is an exec-mask region with its own wait)
        const double v = main_rows[(size_t)min(mr0 + k, n_main_rows - 1) * TC_ICP_SUMS_STRIDE + (threadIdx.x & (TC_ICP_SUMS_STRIDE - 1))];
        fr[k] = (mr0 + k < mr1) ? v : 0.0;
    }
    // refine list = the main blocks' segments; exclusive scan of their counts (every block computes
    // the same scan), then query i -> (segment, local index) by binary search: balanced and
    // deterministic without a global counter
    constexpr int kSegPerRow = kIcpBlock / 64;                         // one segment per wave of a main block
    constexpr int kCntPer = kMaxPartialBlocks * kSegPerRow / kRefineThreads;       // counts per thread at the largest main grid
    const uint32_t n_segs = n_main_rows * kSegPerRow;
    // thread t owns the kCntPer consecutive counts starting at t * kCntPer (a per-lane bound check: a uniform one becomes a
    // branch with the read and its wait inside)
    constexpr uint32_t per = kCntPer;
    uint32_t cnt[kCntPer];
#pragma unroll
    for (int k = 0; k < kCntPer; ++k) {
        const uint32_t b = threadIdx.x * per + k;
        const uint32_t v = rlist[min(b, n_segs - 1)];
        cnt[k] = b < n_segs ? v : 0u;
    }
    if (hd.done) return;
    double folded = 0.0;
#pragma unroll
    for (int k = 0; k < kFoldBatch; ++k) folded += fr[k];              // (rows mr0 .. in order; + 0.0 is exact)
    if (threadIdx.x < TC_ICP_SUMS_STRIDE)
        for (uint32_t r = mr0 + kFoldBatch; r < mr1; ++r) folded += main_rows[(size_t)r * TC_ICP_SUMS_STRIDE + threadIdx.x];   // (never: n_main_rows <= kMaxPartialBlocks)
    const GridGeom &g = tgt.g;
    const float q[4] = {hd.q[0], hd.q[1], hd.q[2], hd.q[3]};          // (GICP: the rotation of the pair terms)
    const float max_dist = hd.max_dist;
    __shared__ uint32_t seg_off[kMaxPartialBlocks * kSegPerRow + 1];
    __shared__ uint32_t wave_tot[kRefineThreads / 64];
    {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kCntPer; ++k) run += cnt[k];
        uint32_t inc = run;                                 // inclusive scan over the block: wave scan + wave totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o);
            if ((int)(threadIdx.x & 63) >= o) inc += v;
        }
        if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t ex = inc - run;
        for (int w2 = 0; w2 < (int)(threadIdx.x >> 6); ++w2) ex += wave_tot[w2];
#pragma unroll
        for (int k = 0; k < kCntPer; ++k) {
            const uint32_t b = threadIdx.x * per + k;
            if (b <= n_segs) seg_off[b] = ex;
            ex += cnt[k];
        }
        if (threadIdx.x == kRefineThreads - 1) seg_off[n_segs] = ex;    // total (n_segs == kCntPer * kRefineThreads: no thread owns slot n_segs)
        __syncthreads();
    }
    const uint32_t count = seg_off[n_segs];
    const uint4 *__restrict__ entries = refine_entries(rlist);
    if (blockIdx.x == 0 && threadIdx.x == 0) {             // statistics
        st->refine_total += count;
        st->refine_max = max(st->refine_max, count);
    }
    if ((uint32_t)blockIdx.x >= count) {                  // no query for this block: folded rows only
        if (threadIdx.x < TC_ICP_SUMS_STRIDE) partial_rows[(size_t)blockIdx.x * TC_ICP_SUMS_STRIDE + threadIdx.x] = folded;
        return;
    }
    const __amdgpu_buffer_rsrc_t pt_rsrc = raw_rsrc(tgt.pts), cs_rsrc = raw_rsrc(tgt.cell_start);
    const int lg = threadIdx.x & (kRG - 1);
    // neighbouring list entries come from one main block = one region of space: deal them out to
    // different refine blocks (group ids are block-minor), or the hard regions land in a few blocks
    const uint32_t group = (threadIdx.x / kRG) * gridDim.x + blockIdx.x, ngroups = gridDim.x * kRefineThreads / kRG;
    const unsigned long long gmask = (kRG == 64) ? ~0ull : (((1ull << kRG) - 1ull) << ((threadIdx.x & 63) & ~(kRG - 1)));
    // per-group f32 sums live in LDS (the group leader is the only writer): no accumulator registers
    // across the search, and the block fold below reads 32 x NACC words instead of reducing 16 waves
    float *const acc = lacc[threadIdx.x / kRG];
    for (int k = lg; k < NACC; k += kRG) acc[k] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // entry of query i: segment = last b with seg_off[b] <= i (kRG-ary search, one pivot per lane of the group)
    auto locate = [&](uint32_t i) -> size_t {
        uint32_t lo = 0, nleft = n_segs;
        while (nleft > 1) {
            const uint32_t stride = (nleft + kRG - 1) / kRG;
            const bool le = (uint32_t)lg * stride < nleft && seg_off[lo + lg * stride] <= i;     // true for a prefix of the lanes
            const uint32_t m = (uint32_t)((__ballot(le) & gmask) >> ((threadIdx.x & 63) & ~(kRG - 1)));
            const uint32_t k = 31u - (uint32_t)__clz((int)m);                                    // lane 0 always holds (invariant)
            lo += k * stride;
            nleft = min(stride, nleft - k * stride);
        }
        return (size_t)lo * seg_stride + (i - seg_off[lo]);
    };
    // the entry of a group's NEXT query is requested before the current one is searched (one round trip less per query)
    uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0;
    if (group < count) { const size_t a = locate(group); n0 = entries[2 * a]; n1 = entries[2 * a + 1]; }
    for (uint32_t i = group; i < count; i += ngroups) {
        const uint4 e0 = n0, e1 = n1;
        if (i + ngroups < count) { const size_t a = locate(i + ngroups); n0 = entries[2 * a]; n1 = entries[2 * a + 1]; }
        const uint32_t j = e0.w, pj = e1.y;
        const float x = __uint_as_float(e0.x), y = __uint_as_float(e0.y), z = __uint_as_float(e0.z);
        const float qx = fminf(fmaxf(x, g.minx), g.maxx), qy = fminf(fmaxf(y, g.miny), g.maxy),
                    qz = fminf(fmaxf(z, g.minz), g.maxz);
        const int cx = cell_coord(qx, g.minx, g.inv_h, g.gx), cy = cell_coord(qy, g.miny, g.inv_h, g.gy),
                  cz = cell_coord(qz, g.minz, g.inv_h, g.gz);
        const float fx = (qx - g.minx) * g.inv_h - (float)cx, fy = (qy - g.miny) * g.inv_h - (float)cy,
                    fz = (qz - g.minz) * g.inv_h - (float)cz;
        const float out2 = outside_d2(x, y, z, qx, qy, qz, g.clamped);
        // start: the best real point the main pass knows (previous match or its ring-1 result, with its distance); ring 1 is done
        unsigned long long bestkey = ~0ull;
        if (pj != 0xFFFFFFFFu) bestkey = ((unsigned long long)e1.x << 32) | pj;
        // A query that arrives WITH a candidate (its previous match or ring-1 best: all but the first iteration's) goes straight to
        // the ball of that distance (round 5): entry -> row windows -> records -> winner, four dependent round trips, where ring 2
        // first and then the ball took six.  The ball includes rings 0 and 1 again -- the same points, the same (distance, position)
        // keys -- which costs reads on an idle chip, not time: the pass 10.0 -> 9.0 us on average (moving phase 11.7 -> 10.4),
        // profiles/r05_refine_pass.txt.  -DTC_REFINE_SHELLS_FIRST: the former order (A/B).
#ifndef TC_REFINE_SHELLS_FIRST
        if (bestkey != ~0ull && !dbg_shells)
            bestkey = refine_ball_scan(tgt, cs_rsrc, pt_rsrc, lg, x, y, z, bestkey, max_dist >= 0.0f ? max_dist * max_dist * 1.0001f : INFINITY);
        else
#endif
        for (int R = 2;; ++R) {
            bool touched;
            const unsigned long long lk = (R == 2) ? refine_shell<2>(tgt, cs_rsrc, pt_rsrc, 2, lg, x, y, z, cx, cy, cz, bestkey, touched)
                                                   : refine_shell<0>(tgt, cs_rsrc, pt_rsrc, R, lg, x, y, z, cx, cy, cz, bestkey, touched);
            bestkey = lk < bestkey ? lk : bestkey;
            const bool any_touched = (__ballot(touched) & gmask) != 0ull;
            const bool covers = (cx - R <= 0) && (cx + R >= g.gx - 1) && (cy - R <= 0) && (cy + R >= g.gy - 1) &&
                                (cz - R <= 0) && (cz + R >= g.gz - 1);
            // clearance to the faces of the ring-R block that have cells beyond them (see nn_search_pruned)
            const float big = 1.0e30f;
            const float mf = fmaxf(fminf(fminf(fminf(cx - R >= 1 ? fx : big, cx + R <= g.gx - 2 ? 1.0f - fx : big),
                                               fminf(cy - R >= 1 ? fy : big, cy + R <= g.gy - 2 ? 1.0f - fy : big)),
                                         fminf(cz - R >= 1 ? fz : big, cz + R <= g.gz - 2 ? 1.0f - fz : big)), 0.0f);
            const float bound = ((float)R + mf - 2e-3f) * g.h;
            const float bd = __uint_as_float((uint32_t)(bestkey >> 32));
#ifdef TC_REFINE_STATS
            if (lg == 0 && (covers || !any_touched || (bestkey != ~0ull && bd <= bound * bound + out2) || (max_dist >= 0.0f && bound > max_dist)))
                atomicAdd(const_cast<uint32_t *>(&st->refine_ring_hist[min(R, 7)]), 1u);
#endif
            if (covers || !any_touched) break;
            if (bestkey != ~0ull && bd <= bound * bound + out2) break;
            if (max_dist >= 0.0f && bound > max_dist) break;   // everything unscanned would be rejected
            // a candidate in hand and the ring rule not met: the rest of its ball, directly (see refine_ball_scan); without a
            // candidate -- an empty neighbourhood -- the shells keep growing until one turns up
            if (bestkey != ~0ull && !(dbg_shells)) {
                bestkey = refine_ball_scan(tgt, cs_rsrc, pt_rsrc, lg, x, y, z, bestkey, max_dist >= 0.0f ? max_dist * max_dist * 1.0001f : INFINITY);
                break;
            }
        }
        if (lg == 0) {
            const float best = __uint_as_float((uint32_t)(bestkey >> 32));
            const uint32_t bestg = (uint32_t)bestkey;
            bool valid = bestkey != ~0ull;
            if (valid && max_dist >= 0.0f) valid = !(sqrtf(best) > max_dist);
            corr_pos[4 * (size_t)j] = valid ? bestg : 0xFFFFFFFFu;
            if (valid) {
                const float4 c = tgt.pts[bestg];
                float a[NACC];
#pragma unroll
                for (int k = 0; k < NACC; ++k) a[k] = acc[k];
                if constexpr (MODE == 2) {
                    float R[9];
                    quat_to_rot(q, R);
                    accumulate_gicp(a, x, y, z, c, best, R, src_cov[2 * (size_t)j], src_cov[2 * (size_t)j + 1], tgt_nrm[2 * (size_t)bestg],
                                    tgt_nrm[2 * (size_t)bestg + 1]);
                } else {
                    float4 n = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (P2PLANE) n = tgt_nrm[bestg];
                    accumulate_pair<P2PLANE, NACC>(g, a, x, y, z, c, n);
                }
#pragma unroll
                for (int k = 0; k < NACC; ++k) acc[k] = a[k];
            }
        }
    }
    // block fold: the groups' rows in group order (f64), plus the folded main rows
    __syncthreads();
    if (threadIdx.x < TC_ICP_SUMS_STRIDE) {
        double sum = 0.0;
        if (threadIdx.x < NACC) {
#pragma unroll 8
            for (int g2 = 0; g2 < kRefineThreads / kRG; ++g2) sum += (double)lacc[g2][threadIdx.x];
        }
        partial_rows[(size_t)blockIdx.x * TC_ICP_SUMS_STRIDE + threadIdx.x] = sum + folded;
    }
    // (Round 4 measured this launch FUSED with icp_finalize -- rows stored write-through, an agent-scope ticket, the last block
    // folds and solves behind sc1 loads, MI355X_MICROARCH.md's hand-off recipe: 12.8 us instead of 4.9 + 4.7 in the aligned phase,
    // 17.9 instead of 16.0 on average; profiles/r04_ab_fused_tail.txt.  The boundary is cheaper than the hand-off.)
}

// ICPResult.correspondences (registration.rs:22-23): matched ORIGINAL target index per ORIGINAL
// source index of the last executed iteration; written once, after the loop.
__global__ void __launch_bounds__(256) icp_write_corr_kernel(GridView tgt, const float4 *__restrict__ src, uint32_t ns,
                                                            const uint32_t *__restrict__ corr_pos,
                                                            uint32_t *__restrict__ corr) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ns) return;
    const uint32_t so = __float_as_uint(src[j].w);
    const uint32_t pj = corr_pos[4 * (size_t)j];           // (stride 4: the w of the working source records)
    corr[so] = (pj == 0xFFFFFFFFu) ? 0xFFFFFFFFu : __float_as_uint(tgt.pts[pj].w);
}

// mse of the last correspondences under the final transform (registration.rs:343-361)
__global__ void __launch_bounds__(kIcpBlock) icp_final_mse_kernel(GridView tgt, const float4 *__restrict__ src, uint32_t ns,
                                                                  uint32_t chunk, const IcpState *__restrict__ st,
                                                                  const uint32_t *__restrict__ corr_pos,
                                                                  double *__restrict__ partials) {
    if (st->done) return;
    __shared__ double red[kIcpBlock / 64][TC_ICP_SUMS_STRIDE];
    const float q[4] = {st->q[0], st->q[1], st->q[2], st->q[3]};
    const float t[3] = {st->t[0], st->t[1], st->t[2]};
    double acc[2] = {0.0, 0.0};
    const uint32_t beg = blockIdx.x * chunk, end = min(beg + chunk, ns);
    for (uint32_t j = beg + threadIdx.x; j < end; j += kIcpBlock) {
        const float4 s = src[j];                             // (the working source record: x, y, z, match)
        const uint32_t bj = __float_as_uint(s.w);
        if (bj == 0xFFFFFFFFu) continue;
        float x, y, z;
        iso_apply(q, t, s.x, s.y, s.z, x, y, z);
        const float4 c = tgt.pts[bj];
        const float ex = x - c.x, ey = y - c.y, ez = z - c.z;
        acc[0] += (double)(ex * ex + ey * ey + ez * ez);
        acc[1] += 1.0;
    }
    block_reduce_store<2>(acc, partials + (size_t)blockIdx.x * TC_ICP_SUMS_STRIDE, red);
}

// vor[p] = 0.25 * (1 - 1e-4) * min(|p - t|^2 over the other target records t of the 3x3x3 block, (0.996 h)^2): a LOWER bound of a
// quarter of the squared distance to the nearest other target point (a record outside the block is at least one cell edge away
// along some axis, also in a clamped grid: the boundary cells are part of the block when they are adjacent).  Squared distances
// computed like every other one here (d2_nc: relative error 4 ulp at most, the subtraction of two floats is exact to half an
// ulp of the DIFFERENCE), so the 1e-4 shave keeps the inscribed-ball test strict.  Duplicate points get 0: never skipped.
__global__ void __launch_bounds__(256) icp_target_nn_bound_kernel(GridView tgt, float4 *__restrict__ vor, const IcpState *__restrict__ st) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const GridGeom &g = tgt.g;
    if (p >= g.n || (st && st->done)) return;
    const float4 q = tgt.pts[p];
    const int cx = cell_coord(q.x, g.minx, g.inv_h, g.gx), cy = cell_coord(q.y, g.miny, g.inv_h, g.gy), cz = cell_coord(q.z, g.minz, g.inv_h, g.gz);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.gx - 1);
    const float hb = 0.996f * g.h;
    float m = hb * hb;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.gz - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.gy - 1); ++y) {
            const uint32_t row = ((uint32_t)z * g.gy + y) * g.gx;
            const uint32_t s = tgt.cell_start[row + x0], e = tgt.cell_start[row + x1 + 1];
            for (uint32_t j = s; j < e; ++j) {
                const float4 c = tgt.pts[j];
                const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
                m = (j != p) ? fminf(m, v) : m;            // fminf ignores a NaN candidate
            }
        }
    vor[p] = make_float4(q.x, q.y, q.z, 0.25f * 0.9999f * m);       // record + bound in one 16-byte gather of the main pass
}

// ---- small f64 solvers (one lane) -----------------------------------------------------------
// 6x6 Cholesky solve, fully unrolled on the packed lower triangle (21 + 6 doubles in registers, static
// indices only, no scratch); rolled loops over LDS arrays measured 8 us slower (every operand an LDS
// round trip on one dependent chain).
// Su = upper triangle by rows (= lower triangle by columns) of the symmetric matrix.
__device__ __forceinline__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }   // r >= c
__device__ __forceinline__ bool chol6_solve(const double *Su, const double *b, double (&x)[6]) {
    double L[21];
    {
        int o = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) { L[tri(c, r)] = Su[o]; ++o; }
    }
    bool ok = true;
    double rinv[6];             // 1 / L[j][j]: the two substitutions multiply by it (twelve f64 divisions less on the serial path)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double f = L[tri(j, k)];
#pragma unroll
            for (int r = j; r < 6; ++r) L[tri(r, j)] -= f * L[tri(r, k)];
        }
        const double d = L[tri(j, j)];
        ok = ok && (d > 0.0);             // Cholesky::new -> None on a non-positive pivot
        // 1 / sqrt(d) from the hardware estimate + two Newton steps (error ~1e-16 relative), sqrt(d) = d / sqrt(d): the correctly
        // rounded sqrt + division this replaces were ~27 dependent f64 instructions per pivot on the ONE lane the whole GPU waits
        // for (round 4: the solve was 2.6 of this launch's 4.7 us)
        const double dp = d > 0.0 ? d : 1.0;
#ifdef TC_OLD_SOLVE          // (A/B: the correctly rounded sqrt + division of rounds 1-3)
        const double sd = sqrt(dp);
        const double inv = 1.0 / sd;
#else
        double inv = __builtin_amdgcn_rsq(dp);
        inv = inv * fma(-0.5 * dp * inv, inv, 1.5);
        inv = inv * fma(-0.5 * dp * inv, inv, 1.5);
        const double sd = dp * inv;
#endif
        rinv[j] = inv;
        L[tri(j, j)] = sd;
#pragma unroll
        for (int r = j + 1; r < 6; ++r) L[tri(r, j)] *= inv;
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[tri(i, k)] * x[k];
        x[i] = s * rinv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[tri(k, i)] * x[k];
        x[i] = s * rinv[i];
    }
    return true;
}

// works in place on LDS (dynamic row indexing in registers would put the whole kernel on scratch)
__device__ __noinline__ bool lu6_solve(double (*A)[6], double *x) {
    for (int i = 0; i < 6; ++i) {
        int piv = i; double best = fabs(A[i][i]);
        for (int r = i + 1; r < 6; ++r) { const double v = fabs(A[r][i]); if (v > best) { best = v; piv = r; } }
        if (A[piv][i] == 0.0) continue;
        if (piv != i) {
            for (int c = 0; c < 6; ++c) { const double tt = A[i][c]; A[i][c] = A[piv][c]; A[piv][c] = tt; }
            const double tt = x[i]; x[i] = x[piv]; x[piv] = tt;
        }
        const double inv = 1.0 / A[i][i];
        for (int r = i + 1; r < 6; ++r) {
            const double f = A[r][i] * inv;
            A[r][i] = f;
            for (int c = i + 1; c < 6; ++c) A[r][c] -= f * A[i][c];
            x[r] -= f * x[i];
        }
    }
    for (int i = 5; i >= 0; --i) {
        if (A[i][i] == 0.0) return false;
        double s = x[i];
        for (int c = i + 1; c < 6; ++c) s -= A[i][c] * x[c];
        x[i] = s / A[i][i];
    }
    return true;
}

__device__ __forceinline__ void quat_mul_f(const float a[4], const float b[4], float o[4]) {
    const float w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    const float i = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const float j = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    const float k = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[0] = i; o[1] = j; o[2] = k; o[3] = w;
}

// current = delta * current  (Isometry3 product, registration.rs:321 / :576); the current transform comes from the header the
// kernel loaded at its start -- reading it again through `st` would be one more round trip to memory on the lane everything waits for
__device__ void compose(const IcpHeader &hd, IcpState *st, const float dq[4], const float dt[3], const GridGeom &g, double searchers, double pairs, int cert_active) {
    // How far this update moves a point x: |R_d x + t_d - x| <= 2 |sin(theta / 2)| |x| + |t_d|, and 2 |sin(theta / 2)| = 2 |(i, j, k)| of
    // the unit delta quaternion.  Left in the state for the next main pass's second-neighbour certificate (icp_correspond_reduce_kernel),
    // which is WANTED once the update is small against the cell edge for a point at the far corner of the target's box -- while
    // the clouds still move by cells nothing would be certified and the bookkeeping would only cost -- AND has something to win: the
    // pass that has just run still sent more than a tenth of its points to a search although the update is small -- sensor noise of the
    // order of the point spacing, which the inscribed-ball test cannot get past (TUM-shaped pair: 85 %).  A clean pair (the uniform
    // benchmark pair once aligned: 0 %) never pays the bookkeeping (one 16-byte read per point and pass, ~2 us per pass at 1 M points).
    // Switching ON also asks for an update three times smaller than staying on does: the uniform benchmark pair hovers around the
    // threshold for twenty iterations while 69 % of its points still search because the clouds still MOVE -- there the wider measuring
    // scans and the bookkeeping cost 2 % of the whole job and certify nothing (measured: 17 440 vs 17 770 it/s).  Once on it stays on
    // while the updates stay small: its own success (0.75 % searchers) must not switch it off.
    // (d_ang >= 0 = WANTED: the finalize launch tells the host -- run_chunked --, which enqueues the certificate's instantiation of the
    // main pass from the chunk after next on; d_run counts the consecutive passes that instantiation has run with small updates: it
    // maintains the bounds from 1 on and uses them from 2 on, and a pass of the plain kernel resets it.)
    // This is one lane's dependent chain at the end of every iteration: the gate is decided before anything is computed, and the
    // lengths come from v_sqrt_f32 (their thresholds carry margins of 1e-4; three correctly rounded square roots were 0.4 us per iteration).
    // Evaluated where somebody reads the answer: by the finalize launch that ends a chunk (bit 1 of cert_active: the host reads the word it
    // leaves) and after every pass of the certificate's instantiation (bit 0); the other iterations keep the state as it is.
    if (cert_active != 0) {
        st->searchers = (uint32_t)fmin(searchers, 4.0e9);
        const bool was_on = hd.d_ang >= 0.0f;
        bool wanted = false;
        float ang = -1.0f, tn = 0.0f;
        if (was_on || searchers > 0.1 * pairs) {
            ang = 2.0f * __builtin_amdgcn_sqrtf(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]) * 1.0001f;
            tn = __builtin_amdgcn_sqrtf(dt[0] * dt[0] + dt[1] * dt[1] + dt[2] * dt[2]) * 1.0001f;
            const float rx_ = fmaxf(fabsf(g.minx), fabsf(g.maxx)), ry_ = fmaxf(fabsf(g.miny), fabsf(g.maxy)), rz_ = fmaxf(fabsf(g.minz), fabsf(g.maxz));
            const float far = __builtin_amdgcn_sqrtf(rx_ * rx_ + ry_ * ry_ + rz_ * rz_) * 1.0001f;
            // (a hundredth of a cell edge to stay on, three thousandths to switch on: the uniform benchmark pair gets below the first after
            // ~40 of its 50 iterations -- and hovers around a tenth of an edge from iteration 15 to 35: tools/dev/delta_probe.py --, the
            // TUM-shaped pair after ~12)
            const float moved = ang * far + tn;
            wanted = ang < 5e-3f && moved < (was_on ? 0.01f : 0.003f) * g.h;
        }
        st->d_ang = wanted ? ang : -1.0f;
        st->d_t = tn;
        st->d_run = (wanted && (cert_active & 1)) ? min(hd.d_run + 1u, 1000000u) : 0u;          // a pass that did not maintain the bounds invalidates them
    }
    const float cq[4] = {hd.q[0], hd.q[1], hd.q[2], hd.q[3]};
    const float ct[3] = {hd.t[0], hd.t[1], hd.t[2]};
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    float rx, ry, rz;
    iso_apply(dq, zero, ct[0], ct[1], ct[2], rx, ry, rz);    // R_delta * t_current
    float nq[4];
    quat_mul_f(dq, cq, nq);
    st->q[0] = nq[0]; st->q[1] = nq[1]; st->q[2] = nq[2]; st->q[3] = nq[3];
    st->t[0] = dt[0] + rx; st->t[1] = dt[1] + ry; st->t[2] = dt[2] + rz;
}

// Kabsch rotation from the 3x3 cross-covariance H = sum p q^T (registration.rs:166-194):
// H = U S V^T by one-sided Jacobi (H V = U S), R = V U^T, reflection fixed on the smallest
// singular direction (the reference negates row 2 of V^T after its descending sort).
__device__ void kabsch_rotation(const double Hin[3][3], double R[3][3]) {
    double H[3][3], V[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { H[i][j] = Hin[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p) for (int qq = p + 1; qq < 3; ++qq) {
            double alpha = 0, beta = 0, gamma = 0;
            for (int i = 0; i < 3; ++i) { alpha += H[i][p] * H[i][p]; beta += H[i][qq] * H[i][qq]; gamma += H[i][p] * H[i][qq]; }
            if (gamma == 0.0 || fabs(gamma) <= 1e-18 * sqrt(alpha * beta)) continue;
            off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
            for (int i = 0; i < 3; ++i) {
                const double hp = H[i][p], hq = H[i][qq];
                H[i][p] = c * hp - s * hq; H[i][qq] = s * hp + c * hq;
                const double vp = V[i][p], vq = V[i][qq];
                V[i][p] = c * vp - s * vq; V[i][qq] = s * vp + c * vq;
            }
        }
        if (off < 1e-15) break;
    }
    double sig[3], U[3][3];
    for (int j = 0; j < 3; ++j) sig[j] = sqrt(H[0][j] * H[0][j] + H[1][j] * H[1][j] + H[2][j] * H[2][j]);
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) if (sig[order[j]] > sig[order[i]]) { int t0 = order[i]; order[i] = order[j]; order[j] = t0; }
    const int i0 = order[0], i1 = order[1], i2 = order[2];
    const double smax = sig[i0];
    for (int i = 0; i < 3; ++i) { U[i][i0] = 0; U[i][i1] = 0; U[i][i2] = 0; }
    if (smax > 0.0) for (int i = 0; i < 3; ++i) U[i][i0] = H[i][i0] / smax; else U[0][i0] = 1.0;
    if (sig[i1] > 1e-14 * smax && sig[i1] > 0.0) {
        for (int i = 0; i < 3; ++i) U[i][i1] = H[i][i1] / sig[i1];
    } else {   // rank 1: any unit vector orthogonal to u0
        double ax = fabs(U[0][i0]), ay = fabs(U[1][i0]), az = fabs(U[2][i0]);
        double e[3] = {0, 0, 0};
        e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0;
        double d = e[0] * U[0][i0] + e[1] * U[1][i0] + e[2] * U[2][i0];
        double v[3] = {e[0] - d * U[0][i0], e[1] - d * U[1][i0], e[2] - d * U[2][i0]};
        double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; ++i) U[i][i1] = v[i] / nv;
    }
    bool u2_from_cross = !(sig[i2] > 1e-14 * smax && sig[i2] > 0.0);
    if (!u2_from_cross) {
        for (int i = 0; i < 3; ++i) U[i][i2] = H[i][i2] / sig[i2];
    } else {
        U[0][i2] = U[1][i0] * U[2][i1] - U[2][i0] * U[1][i1];
        U[1][i2] = U[2][i0] * U[0][i1] - U[0][i0] * U[2][i1];
        U[2][i2] = U[0][i0] * U[1][i1] - U[1][i0] * U[0][i1];
    }
    // R = V U^T
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = V[i][0] * U[j][0] + V[i][1] * U[j][1] + V[i][2] * U[j][2];
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0.0) {   // registration.rs:187-191
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] -= 2.0 * V[i][i2] * U[j][i2];
    }
}

__device__ void rotmat_to_quat(const double m[3][3], float q[4]) {
    const double tr = m[0][0] + m[1][1] + m[2][2];
    double w, i, j, k;
    if (tr > 0.0) {
        const double d = sqrt(tr + 1.0) * 2.0;
        w = 0.25 * d; i = (m[2][1] - m[1][2]) / d; j = (m[0][2] - m[2][0]) / d; k = (m[1][0] - m[0][1]) / d;
    } else if (m[0][0] > m[1][1] && m[0][0] > m[2][2]) {
        const double d = sqrt(1.0 + m[0][0] - m[1][1] - m[2][2]) * 2.0;
        w = (m[2][1] - m[1][2]) / d; i = 0.25 * d; j = (m[0][1] + m[1][0]) / d; k = (m[0][2] + m[2][0]) / d;
    } else if (m[1][1] > m[2][2]) {
        const double d = sqrt(1.0 + m[1][1] - m[0][0] - m[2][2]) * 2.0;
        w = (m[0][2] - m[2][0]) / d; i = (m[0][1] + m[1][0]) / d; j = 0.25 * d; k = (m[1][2] + m[2][1]) / d;
    } else {
        const double d = sqrt(1.0 + m[2][2] - m[0][0] - m[1][1]) * 2.0;
        w = (m[1][0] - m[0][1]) / d; i = (m[0][2] + m[2][0]) / d; j = (m[1][2] + m[2][1]) / d; k = 0.25 * d;
    }
    const double nq = sqrt(w * w + i * i + j * j + k * k);
    q[0] = (float)(i / nq); q[1] = (float)(j / nq); q[2] = (float)(k / nq); q[3] = (float)(w / nq);
}

// convergence bookkeeping shared by both variants (registration.rs:324-339 / :578-592)
__device__ void finish_iteration(const IcpHeader &hd, IcpState *st, float mse, uint32_t n) {
    st->iterations = hd.iterations + 1u;
    st->mse = mse;
    st->n_corr = n;
    const float change = fabsf(hd.prev_mse - mse);
    if (change < hd.conv_thr) { st->converged = 1; st->done = 1; return; }
    st->prev_mse = mse;
}

constexpr int kFinalizeThreads = 512;
// the calling block (kFinalizeThreads threads): fixed-order sum of the rows, solve, compose, bookkeeping
template <int MODE>
__device__ void finalize_body(const double *__restrict__ partials, uint32_t nblocks, IcpState *__restrict__ st, const GridGeom &g,
                              int do_sum, int do_apply, double (*sm)[TC_ICP_SUMS_STRIDE], const IcpHeader &hd, int cert_active = 0) {
    const bool done = hd.done != 0;
    constexpr bool P2PLANE = MODE != 0;        // GICP solves the same 6x6 system from the same 29 words (gicp.rs:258-281)
    if (do_sum) {
        // 16 row groups x 32 columns (512 threads): every group folds its rows in a fixed order, then column t folds the 16
        // groups in order.  The usual case (kRefineBlocks = 256 rows) is ONE round of 16 loads in flight per thread (it was two
        // rounds with 8 groups: the pass is the latency of its rounds)
        constexpr int G = kFinalizeThreads / 32;
        {
            const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            uint32_t b = grp;
            if (nblocks == 256) {
                double v[256 / G];
#pragma unroll
                for (int k = 0; k < 256 / G; ++k) v[k] = partials[(size_t)(grp + G * k) * TC_ICP_SUMS_STRIDE + col];
#pragma unroll
                for (int k = 0; k < 256 / G; k += 4) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; s3 += v[k + 3]; }
                b = 256 + grp;
            }
            for (; b + 3 * G < nblocks; b += 4 * G) {
                s0 += partials[(size_t)b * TC_ICP_SUMS_STRIDE + col];
                s1 += partials[(size_t)(b + G) * TC_ICP_SUMS_STRIDE + col];
                s2 += partials[(size_t)(b + 2 * G) * TC_ICP_SUMS_STRIDE + col];
                s3 += partials[(size_t)(b + 3 * G) * TC_ICP_SUMS_STRIDE + col];
            }
            for (; b < nblocks; b += G) s0 += partials[(size_t)b * TC_ICP_SUMS_STRIDE + col];
            sm[grp][col] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        double tot = 0.0;
        if (threadIdx.x < TC_ICP_SUMS_STRIDE) {
#pragma unroll
            for (int gi = 0; gi < G; ++gi) tot += sm[gi][threadIdx.x];
            if (!done) st->sums[threadIdx.x] = tot;
        }
        __syncthreads();
        if (threadIdx.x < TC_ICP_SUMS_STRIDE) sm[0][threadIdx.x] = tot;      // the solve reads them from LDS, not back from memory
        __syncthreads();
    }
    if (!do_apply || threadIdx.x != 0 || done) return;
    const double *S = do_sum ? &sm[0][0] : st->sums;
    if (P2PLANE) {
        const double cnt = S[28];
        if (cnt < 6.0) { st->status = TC_ALGORITHM; st->done = 1; return; }     // registration.rs:568-572
        double x[6];
        if (!chol6_solve(S, S + 21, x)) {
            // LU fallback on LDS arrays (dynamic row indexing; rare)
            __shared__ double sA[6][6], sx[6];
            int o = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c) { sA[r][c] = S[o]; sA[c][r] = S[o]; ++o; }
#pragma unroll
            for (int r = 0; r < 6; ++r) sx[r] = S[21 + r];
            if (!lu6_solve(sA, sx)) { st->status = TC_ALGORITHM; st->done = 1; return; }   // :432-438
#pragma unroll
            for (int r = 0; r < 6; ++r) x[r] = sx[r];
        }
        // Rz(x2) * Ry(x1) * Rx(x0) as axis-angle quaternions (:441-444), f32 like the reference
        const float hx = (float)x[0] / 2.0f, hy = (float)x[1] / 2.0f, hz = (float)x[2] / 2.0f;
        float sx, cx, sy, cy, sz, cz;          // (one argument reduction per angle)
        sincosf(hx, &sx, &cx); sincosf(hy, &sy, &cy); sincosf(hz, &sz, &cz);
        const float qx[4] = {sx, 0.0f, 0.0f, cx};
        const float qy[4] = {0.0f, sy, 0.0f, cy};
        const float qz[4] = {0.0f, 0.0f, sz, cz};
        float zy[4], rot[4];
        quat_mul_f(qz, qy, zy);
        quat_mul_f(zy, qx, rot);
        const float dt[3] = {(float)x[3], (float)x[4], (float)x[5]};
        compose(hd, st, rot, dt, g, S[kSearchersCol], cnt, cert_active);
        finish_iteration(hd, st, (float)(S[27] / cnt), (uint32_t)cnt);
    } else {
        const double cnt = S[16];
        if (cnt < 3.0) { st->status = TC_ALGORITHM; st->done = 1; return; }     // registration.rs:311-315
        const double ms[3] = {S[0] / cnt, S[1] / cnt, S[2] / cnt}, mq[3] = {S[3] / cnt, S[4] / cnt, S[5] / cnt};
        double H[3][3], R[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r][c] = S[6 + 3 * r + c] - cnt * ms[r] * mq[c];
        if (st->kiss) {         // svd_transform: |H|_F < 1e-10 -> Algorithm (kiss_icp.rs:130-136)
            double hn = 0.0;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) hn += H[r][c] * H[r][c];
            if (sqrt(hn) < 1e-10) { st->status = TC_ALGORITHM; st->done = 1; return; }
        }
        kabsch_rotation(H, R);
        const double cs[3] = {ms[0] + (double)g.cx, ms[1] + (double)g.cy, ms[2] + (double)g.cz};
        const double cq[3] = {mq[0] + (double)g.cx, mq[1] + (double)g.cy, mq[2] + (double)g.cz};
        float dq[4];
        rotmat_to_quat(R, dq);
        float dt[3];
        for (int r = 0; r < 3; ++r) dt[r] = (float)(cq[r] - (R[r][0] * cs[0] + R[r][1] * cs[1] + R[r][2] * cs[2]));
        compose(hd, st, dq, dt, g, S[kSearchersCol], cnt, cert_active);
        double nmse = S[15];                               // sum |s - q|^2 before the update (registration.rs:214)
        if (st->kiss) {
            // KISS-ICP measures AFTER applying delta (kiss_icp.rs:270-276).  With the optimal translation the
            // residual is R (s - c_s) - (q - c_q):  sum = sum|s-q|^2 - n |c_s - c_q|^2 + 2 tr H - 2 sum_ij R_ij H_ji
            const double d0 = ms[0] - mq[0], d1 = ms[1] - mq[1], d2 = ms[2] - mq[2];
            double rh = 0.0;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rh += R[i][j] * H[j][i];
            nmse = S[15] - cnt * (d0 * d0 + d1 * d1 + d2 * d2) + 2.0 * (H[0][0] + H[1][1] + H[2][2]) - 2.0 * rh;
            nmse = fmax(nmse, 0.0);
        }
        finish_iteration(hd, st, (float)(nmse / cnt), (uint32_t)cnt);
    }
}

template <int MODE>
// done_out: when set (the last iteration of a chunk of the host's enqueue schedule), the `done` flag is also written there --
// a word of the context's pinned, device-visible host block, which the host reads after the chunk's event: no copy kernel on
// the stream between two iterations (a 4-byte hipMemcpyAsync is a 4.7 us blit kernel; eight of them per 50-iteration call)
__global__ void __launch_bounds__(kFinalizeThreads) icp_finalize_kernel(const double *__restrict__ partials, uint32_t nblocks,
                                                           IcpState *__restrict__ st, const GridGeom g, int do_sum, int do_apply,
                                                           int32_t *__restrict__ done_out, int cert_active) {
    // (the rows are requested before `done` is known: the test is one more round trip to memory in front of them otherwise)
    const IcpHeader hd = load_header(st);
    __shared__ double sm[kFinalizeThreads / 32][TC_ICP_SUMS_STRIDE];
    // (done_out: 1 = the registration is over, 2 = this chunk of the host's enqueue schedule has run and it is not; the host
    // polls the word, which lives in pinned host memory: run_chunked)
    if (hd.done && !do_sum) { if (done_out && threadIdx.x == 0) __hip_atomic_store(done_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
#ifdef TC_PHASE_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    finalize_body<MODE>(partials, nblocks, st, g, do_sum, 0, sm, hd);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    finalize_body<MODE>(partials, nblocks, st, g, 0, do_apply, sm, hd, cert_active | (done_out != nullptr ? 2 : 0));
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { st->refine_ring_hist[0] += (uint32_t)(t1 - t0); st->refine_ring_hist[1] += (uint32_t)(t2 - t1); }
#else
    finalize_body<MODE>(partials, nblocks, st, g, do_sum, do_apply, sm, hd, cert_active | (done_out != nullptr ? 2 : 0));
#endif
    if (done_out && threadIdx.x == 0)      // (thread 0 wrote st->done itself)
        // (3 = the chunk has run, the registration goes on AND wants the second-neighbour certificate: compose(), run_chunked)
        __hip_atomic_store(done_out, (hd.done || st->done) ? 1 : (st->d_ang >= 0.0f ? 3 : 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// after the last iteration: not-converged epilogue (registration.rs:343-369 / :595-601)
// from_sums: the post-loop (sum |T s - q|^2, count) of a sharded point-to-point run sit in st->sums[0..1], already
// reduced over the ranks (icp_final_mse_fold_kernel + all-reduce); otherwise the per-block rows are folded here
// host_out (optional): the pinned host block's copy of the state -- everything in front of `sums` is stored there directly (system
// scope; the host reads it after the call's final stream synchronisation): no device-to-host copy kernel between this launch and
// the correspondence write-out
constexpr int kStateWords = (int)(offsetof(IcpState, sums) / sizeof(uint32_t));
static_assert(kStateWords <= 64, "one lane per word");
// called by the whole (one-wave) block: lane 0, which wrote the state, parks its words in LDS (a lane reads its OWN stores back in
// order), the lanes store one word each
__device__ __forceinline__ void publish_state(const IcpState *st, IcpState *host_out, uint32_t *lds) {
    if (threadIdx.x == 0) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(st);
        for (int i = 0; i < kStateWords; ++i) lds[i] = s[i];
    }
    __syncthreads();
    if ((int)threadIdx.x < kStateWords)
        __hip_atomic_store(reinterpret_cast<uint32_t *>(host_out) + threadIdx.x, lds[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void icp_finish_kernel(IcpState *__restrict__ st, const double *__restrict__ partials, uint32_t nblocks, int p2plane, int from_sums,
                                  IcpState *__restrict__ host_out) {
    __shared__ uint32_t lds[64];
    if (threadIdx.x == 0 && !st->done) {
    st->converged = 0;
    if (p2plane || st->kiss) {         // point-to-plane :595-601 and kiss_icp.rs:292-299 return the last measured mse
        st->mse = st->prev_mse;
    } else {
        double s = 0.0, c = 0.0;
        if (from_sums) { s = st->sums[0]; c = st->sums[1]; }
        else for (uint32_t b = 0; b < nblocks; ++b) { s += partials[(size_t)b * TC_ICP_SUMS_STRIDE]; c += partials[(size_t)b * TC_ICP_SUMS_STRIDE + 1]; }
        st->mse = (c > 0.0) ? (float)(s / c) : st->prev_mse;
    }
    st->done = 1;
    }
    if (host_out) publish_state(st, host_out, lds);
}

// sharded point-to-point: this rank's post-loop (sum, count) -> st->sums[0..1] (rest zero), same order as icp_finish_kernel
__global__ void icp_final_mse_fold_kernel(IcpState *__restrict__ st, const double *__restrict__ partials, uint32_t nblocks) {
    if (threadIdx.x >= TC_ICP_SUMS_STRIDE) return;
    double s = 0.0;
    if (threadIdx.x < 2 && !st->done)
        for (uint32_t b = 0; b < nblocks; ++b) s += partials[(size_t)b * TC_ICP_SUMS_STRIDE + threadIdx.x];
    st->sums[threadIdx.x] = s;
}

// ---- host orchestration ---------------------------------------------------------------------
struct IcpLaunch {
    uint32_t nblocks = 0, chunk = 0;          // correspond_reduce grid (= rows of partials) and points per block
    uint32_t mse_blocks = 0, mse_chunk = 0;   // final-mse grid (p2p)
};
static IcpLaunch plan_launch(size_t ns) {
    IcpLaunch l;
    // ONE round of blocks: 4 blocks of 256 lanes per CU = 1024 resident blocks on 256 CUs; 1 M points: 977 blocks of
    // 1024 points (4 per lane).  Measured alternatives at 1 M points (45 us): 6 blocks per CU x 768 points (80 VGPRs,
    // 6 waves per SIMD, still one round): 48 us -- more waves do not pay for the extra per-block sums; any grid that
    // needs a second round of blocks (e.g. 1303 blocks at 5 per CU): 54-62 us, the tail.
#ifndef TC_ICP_RESIDENT
#define TC_ICP_RESIDENT 4
#endif
    constexpr size_t kResidentBlocks = TC_ICP_RESIDENT * 256;      // (-DTC_ICP_RESIDENT=3 -DTC_ICP_LDS_PAD=..: the occupancy probe of profiles/r05_main_pass_occupancy.txt)
    size_t chunk = (ns + kResidentBlocks - 1) / kResidentBlocks;
    chunk = std::max<size_t>((chunk + kIcpBlock - 1) / kIcpBlock * kIcpBlock, kIcpBlock);
    uint32_t nb = (uint32_t)((ns + chunk - 1) / chunk);
    nb = std::max<uint32_t>((nb + 7) / 8 * 8, 8);
    l.nblocks = nb; l.chunk = (uint32_t)chunk;
    l.mse_blocks = nb; l.mse_chunk = (uint32_t)chunk;
    return l;
}

// tile dims for the source ordering: compact blocks of ~256 source points
static TileGeom plan_tiles(const GridGeom &g, size_t ns) {
    const double rho = (double)ns / std::max<double>(g.ncell, 1.0);
    const double want = 256.0 / std::max(rho, 1e-6);      // cells per tile
    static const int cand[][3] = {{8, 2, 2}, {8, 3, 2}, {8, 3, 3}, {8, 4, 3}, {8, 4, 4}, {8, 5, 4}, {8, 5, 5}, {8, 6, 5},
                                  {8, 6, 6}, {10, 6, 6}, {12, 6, 6}, {16, 6, 6}, {16, 8, 8}, {4, 2, 2}, {4, 2, 1}, {2, 2, 1}};
    if (const char *e = getenv("TC_ICP_TILE")) {          // experiments: "tx,ty,tz"
        int a = 0, b = 0, c = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) return make_tiles(g, a, b, c);
    }
    int best = 0; double bd = 1e300;
    for (int i = 0; i < (int)(sizeof(cand) / sizeof(cand[0])); ++i) {
        const double c = (double)cand[i][0] * cand[i][1] * cand[i][2];
        const double d = std::fabs(std::log(c / want));
        if (d < bd) { bd = d; best = i; }
    }
    return make_tiles(g, cand[best][0], cand[best][1], cand[best][2]);
}

// mode: 0 point-to-point, 1 point-to-plane (nrm = target normals), 2 GICP (nrm = target covariances, src_cov)
static void launch_iteration(tc_context *ctx, int mode, const GridView &tv, const float4 *nrm, const float4 *src,
                             uint32_t ns, const IcpLaunch &l, IcpState *st, uint32_t *corr_pos, uint32_t *rlist,
                             double *partials, bool do_sum, bool do_apply, bool do_reduce, const float4 *src_cov = nullptr,
                             const float4 *vor = nullptr, int32_t *done_out = nullptr, float4 *wl = nullptr, bool cert = false) {
    hipStream_t s = ctx->stream;
    const int dbg = debug_flags() | (ctx->profiling == 3 ? 8 : 0);
    double *refine_rows = partials + (size_t)l.nblocks * TC_ICP_SUMS_STRIDE;
    if (do_reduce) {
        {
            ProfScope ps(ctx, mode == 1 ? "icp_correspond_reduce_p2plane" : mode == 2 ? "icp_correspond_reduce_gicp" : "icp_correspond_reduce_p2p", true);
            auto kern = mode == 1 ? icp_correspond_reduce_kernel<1> : mode == 2 ? icp_correspond_reduce_kernel<2> : icp_correspond_reduce_kernel<0>;
            if (dbg & 8) kern = mode == 1 ? icp_correspond_reduce_kernel<1, true> : mode == 2 ? icp_correspond_reduce_kernel<2, true> : icp_correspond_reduce_kernel<0, true>;
            // the certificate's instantiation (point-to-point and point-to-plane; the counting instantiation and GICP stay plain)
            cert = cert && wl != nullptr && mode != 2 && !(dbg & 8);
            if (cert) kern = mode == 1 ? icp_correspond_reduce_kernel<1, false, true> : icp_correspond_reduce_kernel<0, false, true>;
            // a chunk's last main pass counts its searching lanes for the gate its finalize launch evaluates (point-to-point / point-to-plane)
            else if (done_out != nullptr && wl != nullptr && mode != 2 && !(dbg & 8))
                kern = mode == 1 ? icp_correspond_reduce_kernel<1, false, false, true> : icp_correspond_reduce_kernel<0, false, false, true>;
            const float4 *vor_arg = (dbg & 4) ? nullptr : vor;
            unsigned long long *times_arg = (dbg & 1024) ? (unsigned long long *)ctx->dbg_times.p : nullptr;
            if (ps.active())          // a timed launch: the events carry the kernel's own start / end stamps (ProfScope)
                hipExtLaunchKernelGGL(kern, dim3(l.nblocks), dim3(kIcpBlock), 0, s, ps.e0, ps.e1, 0, tv, nrm, const_cast<float4 *>(src), ns, l.chunk,
                                      (const IcpState *)st, rlist, partials, dbg, src_cov, vor_arg, times_arg, tv.pts12);
            else
                hipLaunchKernelGGL(kern, dim3(l.nblocks), dim3(kIcpBlock), 0, s, tv, nrm, const_cast<float4 *>(src), ns, l.chunk, st, rlist, partials, dbg, src_cov,
                                   vor_arg, times_arg, tv.pts12);
        }
        ProfScope ps(ctx, "icp_refine");
        auto kern = mode == 1 ? icp_refine_kernel<1> : mode == 2 ? icp_refine_kernel<2> : icp_refine_kernel<0>;
        hipLaunchKernelGGL(kern, dim3(kRefineBlocks), dim3(kRefineThreads), 0, s, tv, nrm, src, st, corr_pos, rlist, partials, l.nblocks, l.chunk / (kIcpBlock / 64),
                           refine_rows, src_cov, dbg);
    }
    if (do_sum || do_apply) {
        ProfScope ps(ctx, "icp_finalize");
        const uint32_t rows = kRefineBlocks;      // the refine pass folded the main pass's rows into its own
        auto kern = mode == 0 ? icp_finalize_kernel<0> : icp_finalize_kernel<1>;
        hipLaunchKernelGGL(kern, dim3(1), dim3(kFinalizeThreads), 0, s, refine_rows, rows, st, tv.g, do_sum ? 1 : 0, (do_apply && !(dbg & 32)) ? 1 : 0, do_apply ? done_out : nullptr, (cert && do_reduce) ? 1 : 0);
    }
}

// The inscribed-ball bounds of the target (icp_target_nn_bound_kernel) cost about one iteration's main pass: they are computed
// when a registration is still running after 6 iterations (TC_VOR_AFTER overrides, for experiments) (scan-to-scan odometry that converges in a handful of
// iterations never pays for them).  Exact either way: with or without them every iteration finds the same matches.
static size_t vor_after() {
    static const size_t v = [] { const char *e = getenv("TC_VOR_AFTER"); return e ? (size_t)atoi(e) : (size_t)6; }();
    return v;
}
static tc_status launch_target_nn_bounds(tc_context *ctx, DeviceIndex &ix, const GridView &tv, const IcpState *st, const float4 **out) {
    *out = nullptr;
    if (ix.vor_valid) { *out = (const float4 *)ix.vor.p; return TC_OK; }      // a cloud handle that has been a target before
    if (tc_status s = ensure(ctx, ix.vor, (size_t)ix.geom.n * sizeof(float4))) return s;
    ProfScope ps(ctx, "icp_target_nn_bounds");
    // (st == nullptr: unconditional; otherwise skipped once the registration is done)
    const bool persistent = &ix != &ctx->tgt_index;          // a cloud handle's index: the bounds are kept for its later registrations
    hipLaunchKernelGGL(icp_target_nn_bound_kernel, dim3((ix.geom.n + 255) / 256), dim3(256), 0, ctx->stream, tv, (float4 *)ix.vor.p,
                       persistent ? nullptr : st);
    ix.vor_valid = persistent;
    *out = (const float4 *)ix.vor.p;
    return TC_OK;
}

// ~1.45 pts/cell.  Scanned again after the main / refine split (50-iteration ICP, 1 M points): 0.8 -> 6.55 ms, 0.9 -> 5.55,
// 1.0 -> 5.00, 1.13 -> 4.75, 1.25 -> 4.72, 1.4 -> 4.72 (flat: the main pass grows as the refine pass shrinks)
float icp_cell_factor() {             // (TC_ICP_CELL_FACTOR: tuning experiments)
    static const float v = [] { const char *e = getenv("TC_ICP_CELL_FACTOR"); return e ? (float)atof(e) : 1.13f; }();
    return v;
}   // ~1.45 pts/cell: ring 1 is exact for ~99.8 % of uniform queries

// the target's sorted records again as packed 12-byte x, y, z (the candidate loop's array), padded like the records (+inf-like
// coordinates behind the last one: a step reads up to three records past its span)
__global__ void __launch_bounds__(256) icp_pack12_kernel(const float4 *__restrict__ pts, uint32_t n, uint32_t npad, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    float4 r = make_float4(__uint_as_float(0x7F7F7F7Fu), __uint_as_float(0x7F7F7F7Fu), __uint_as_float(0x7F7F7F7Fu), 0.f);
    if (i < n) r = pts[i];
    out[3 * (size_t)i] = r.x; out[3 * (size_t)i + 1] = r.y; out[3 * (size_t)i + 2] = r.z;
}

// the loop's working copy of the ordered source: x, y, z and, in w, the position of the point's current match (none yet)
__global__ void __launch_bounds__(256) icp_working_source_kernel(const float4 *__restrict__ src, uint32_t n, float4 *__restrict__ out, float4 *__restrict__ wl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (wl) wl[i] = make_float4(0.f, 0.f, 0.f, 0.f);           // (no second-neighbour bound yet)
    float4 r = src[i];
    r.w = __uint_as_float(0xFFFFFFFFu);
    out[i] = r;
}

struct IcpSetup {
    IcpLaunch l;
    GridView tv;
    TileGeom tg;
    DeviceIndex *tix = nullptr;      // the target's index: ctx->tgt_index, or a cloud handle's
    const float4 *src = nullptr;     // the source records in the order the loop walks them (w = original index)
    float4 *wsrc = nullptr;          // the loop's working copy of them (w = position of the current match): what the kernels read
    float4 *wl = nullptr;            // per source point: reference position + second-neighbour bound of the main pass (behind the working copy)
};
// the matches as the refine / write-out kernels address them: the w of the working records, stride 4
static inline uint32_t *match_words(float4 *wsrc) { return reinterpret_cast<uint32_t *>(wsrc) + 3; }

static tc_status icp_setup(tc_context *ctx, bool p2plane, const float *d_src, size_t ns, const float *d_tgt, size_t nt,
                           const float *d_nrm, size_t nstride, const float init[7], float max_dist, float conv_thr,
                           IcpSetup &out, int kiss = 0, DeviceIndex *tgt_prebuilt = nullptr, const DeviceIndex *src_presorted = nullptr,
                           bool strict_source_order = false) {
    // the search addresses target records by 32-bit byte offsets (16 B each)
    if (nt >= (1ull << 28) || ns >= (1ull << 28)) return fail(ctx, TC_UNSUPPORTED, "ICP clouds are limited to 2^28 - 1 points");
    out.tix = tgt_prebuilt ? tgt_prebuilt : &ctx->tgt_index;
    if (!tgt_prebuilt) {
        if (tc_status s = build_index(ctx, ctx->tgt_index, d_tgt, nt, icp_cell_factor(), nullptr, nullptr, nullptr, 0.0f, 2.5f)) return s;
        if (tc_status s = wait_uploads(ctx)) return s;          // a host entry point's source / normals, uploaded under the build
        if (p2plane)
            if (tc_status s = gather_normals(ctx, ctx->tgt_index, d_nrm, nstride)) return s;
    }
    if (tc_status s = wait_uploads(ctx)) return s;
    if (tc_status s = ensure(ctx, ctx->state, sizeof(IcpState))) return s;
    IcpState *hs = (IcpState *)((char *)ctx->pinned + 256);
    std::memset(hs, 0, sizeof(IcpState));
    for (int i = 0; i < 4; ++i) hs->q[i] = init[i];
    for (int i = 0; i < 3; ++i) hs->t[i] = init[4 + i];
    hs->prev_mse = INFINITY;
    hs->d_ang = -1.0f; hs->d_t = 0.0f; hs->d_run = 0;
    hs->conv_thr = conv_thr;
    hs->max_dist = max_dist;
    hs->kiss = kiss;
    hs->status = TC_OK;
    TC_HIP_TRY(ctx, hipMemcpyAsync(ctx->state.p, hs, sizeof(IcpState), hipMemcpyHostToDevice, ctx->stream));
    // order the source by the (tile-major) target cell of its initially transformed position -- or take the order a source
    // HANDLE already has from its own index (cell-sorted in its own grid: spatially coherent as well; a frame of a stream is
    // indexed once, for its normals and as the next registration's target, and not a second time as this one's source)
    out.tg = plan_tiles((*out.tix).geom, ns);
    if (src_presorted && src_presorted->geom.n == ns) {
        out.src = (const float4 *)src_presorted->pts.p;
    } else {
        if (ns > 0)       // (a rank of a sharded run may own no source points)
            if (tc_status s = build_index(ctx, ctx->src_index, d_src, ns, 0.0f, &(*out.tix).geom, (const IcpState *)ctx->state.p, &out.tg, 0.0f, 0.0f,
                                          strict_source_order)) return s;
        out.src = (const float4 *)ctx->src_index.pts.p;
    }
    out.l = plan_launch(ns);
    if (tc_status s = ensure(ctx, ctx->partials, ((size_t)(kMaxPartialBlocks + kRefineBlocks) * TC_ICP_SUMS_STRIDE + 2) * sizeof(double))) return s;
    // corr | corr_pos | refine counts (one per wave of a main block) | refine entries (32 bytes each, chunk / 4 per wave)
    if (tc_status s = ensure(ctx, ctx->corr, (2 * ns + (size_t)kMaxPartialBlocks * (kIcpBlock / 64) + 4 + kRefineEntryWords * (size_t)out.l.nblocks * out.l.chunk) * sizeof(uint32_t))) return s;
    if (!out.tix->pts12_valid) {          // once per indexed cloud (a handle keeps it for its later registrations)
        const uint32_t npad = (uint32_t)nt + 16u;
        if (tc_status s = ensure(ctx, out.tix->pts12, (size_t)npad * 12 + 64)) return s;
        ProfScope ps(ctx, "icp_pack12");
        hipLaunchKernelGGL(icp_pack12_kernel, dim3((npad + 255) / 256), dim3(256), 0, ctx->stream, (const float4 *)(*out.tix).pts.p, (uint32_t)nt, npad,
                           (float *)out.tix->pts12.p);
        out.tix->pts12_valid = true;
    }
    out.tv = view_of((*out.tix));
    if (tc_status s = ensure(ctx, ctx->icp_wsrc, 2 * (ns + 4) * sizeof(float4))) return s;
    out.wsrc = (float4 *)ctx->icp_wsrc.p;
    out.wl = out.wsrc + ns + 4;
    if (ns > 0) {
        ProfScope ps(ctx, "icp_working_source");
        hipLaunchKernelGGL(icp_working_source_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, out.src, (uint32_t)ns, out.wsrc, (float4 *)nullptr);
    }
    return TC_OK;
}

// per-point covariances (8 floats per point in ORIGINAL order: xx xy xz yy yz zz - -) into the order of the
// sorted records, two float4 per position
__global__ void __launch_bounds__(256) gather_cov_kernel(const float4 *__restrict__ pts_sorted, uint32_t n, const float4 *__restrict__ cov_orig,
                                                        float4 *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = __float_as_uint(pts_sorted[i].w);
    out[2 * (size_t)i] = cov_orig[2 * (size_t)o];
    out[2 * (size_t)i + 1] = cov_orig[2 * (size_t)o + 1];
}

static tc_status icp_run_mode(tc_context *ctx, int mode, const float *d_src, size_t ns, const float *d_tgt, size_t nt,
                              const float *d_nrm, size_t nstride, const float init[7], size_t max_iters, float max_dist,
                              float conv_thr, tc_icp_result *res, bool corr_on_device, int kiss, const float *d_cov_src,
                              const float *d_cov_tgt, DeviceIndex *tgt_prebuilt = nullptr, const DeviceIndex *src_presorted = nullptr);

tc_status icp_run(tc_context *ctx, bool p2plane, const float *d_src, size_t ns, const float *d_tgt, size_t nt,
                  const float *d_nrm, size_t nstride, const float init[7], size_t max_iters, float max_dist,
                  float conv_thr, tc_icp_result *res, bool corr_on_device, int kiss, DeviceIndex *tgt_prebuilt, const DeviceIndex *src_presorted) {
    return icp_run_mode(ctx, p2plane ? 1 : 0, d_src, ns, d_tgt, nt, d_nrm, nstride, init, max_iters, max_dist, conv_thr, res, corr_on_device,
                        kiss, nullptr, nullptr, tgt_prebuilt, src_presorted);
}

// gicp.rs:157-305 once the covariances exist (d_cov_*: 8 floats per point, original order)
tc_status icp_run_gicp(tc_context *ctx, const float *d_src, size_t ns, const float *d_tgt, size_t nt, const float *d_cov_src,
                       const float *d_cov_tgt, const float init[7], size_t max_iters, float max_dist, float conv_thr,
                       tc_icp_result *res, bool corr_on_device) {
    return icp_run_mode(ctx, 2, d_src, ns, d_tgt, nt, nullptr, 0, init, max_iters, max_dist, conv_thr, res, corr_on_device, 0, d_cov_src,
                        d_cov_tgt);
}

// A run that did not converge has executed exactly max_iters iterations (registration.rs:278 / :533): the device counts them
// Counters of the main pass's counting instantiation (tc_profile_enable(ctx, 3) / TC_DEBUG & 8), summed over the call's iterations, into
// the context's 64-bit session totals.  Called by every road that runs the main pass: plain / handle calls, the sharded loop, the
// shard handles (the sharded roads used to drop them).  Slots of IcpState::refine_ring_hist: [2] steps the trips took,
// [3..4] steps the lanes needed (one 64-bit word), [5] wave trips, [6] trips without a search, [7] searches.  (A -DTC_REFINE_STATS
// build uses the same array as its exit-ring histogram and -DTC_PHASE_STAMPS words 0 / 1: development builds, never together with mode 3.)
static void fold_search_stats(tc_context *ctx, const IcpState *hs) {
    if (!((debug_flags() & 8) || ctx->profiling == 3)) return;
    const uint32_t *h = hs->refine_ring_hist;
    unsigned long long needed;
    std::memcpy(&needed, &h[3], sizeof(needed));
    ctx->stat_icp[0] += hs->iterations; ctx->stat_icp[1] += h[5]; ctx->stat_icp[2] += h[6]; ctx->stat_icp[3] += h[7];
    ctx->stat_icp[4] += needed; ctx->stat_icp[5] += h[2];
    if (debug_flags() & 8)
        fprintf(stderr, "[tc] icp main pass over %u iterations: wave trips %u, without a search %u (%.1f %%); searches %u, candidate steps needed %llu "
                        "(%.2f per search), steps taken by the trips' slowest lanes %u (%.2f per searching trip): lock-step ratio (lane slots spent / steps needed) %.2f\n",
                hs->iterations, h[5], h[6], 100.0 * h[6] / std::max(h[5], 1u), h[7], needed, (double)needed / std::max(h[7], 1u), h[2],
                (double)h[2] / std::max(h[5] - h[6], 1u), (double)h[2] * 64.0 / std::max<double>((double)needed, 1.0));
}

static tc_status check_iteration_count(tc_context *ctx, const IcpState *hs, size_t max_iters) {
    if (hs->status == TC_OK && !hs->converged && hs->iterations != max_iters)
        return fail(ctx, TC_GPU, "internal error: the ICP loop executed " + std::to_string(hs->iterations) + " of " + std::to_string(max_iters) + " iterations");
    return TC_OK;
}

// Iterations are enqueued in chunks; the `done` flag of chunk c is polled (the chunk's last finalize launch writes it into the
// pinned host block; event) before chunk c + 2 is enqueued, so the stream never drains while running.  Two chunks are always in flight and chunk c + 2 is only enqueued
// when chunk c did not finish the job: a first chunk of 6 and a second of 2 make a registration that converges within 6
// iterations -- scan-to-scan odometry -- pay 8 iterations of launches instead of 16 (that was 1/3 of a LiDAR frame's time).
// The events live in the context (created once, reused by every call).
template <typename F>
static tc_status run_chunked(tc_context *ctx, size_t max_iters, IcpState *dstate, F &&enqueue_iteration) {
    // (round 6: two chunks of 4 before the 8s -- the certificate's instantiation of the main pass starts two chunks after the finalize
    // launch that asks for it, iteration 17 instead of 21 on the TUM-shaped pair)
    auto chunk_len = [](size_t c) -> size_t { return c == 0 ? 6 : c == 1 ? 2 : c <= 4 ? 4 : 8; };
    size_t nchunks = 0;
    for (size_t covered = 0; covered < max_iters; ++nchunks) covered += chunk_len(nchunks);
    // The last launch of chunk c writes 1 (registration over) or 2 (chunk over, registration not) into word c of the pinned,
    // device-visible block, and the host POLLS that word: no event on the stream (a recorded event is a 5.5 us bubble between two
    // iterations, eight of them per 50-iteration call: round 4).
    volatile int32_t *flags = (volatile int32_t *)((char *)ctx->pinned + 1024);
    int32_t *d_flags = nullptr;                                    // the same words as the device sees them
    TC_HIP_TRY(ctx, hipHostGetDevicePointer((void **)&d_flags, (void *)flags, 0));
    const size_t max_flags = 200;                                  // pinned bytes 1024 .. 2048 hold them (the bbox partials follow)
    size_t it = 0;
    // (a registration starts with the instantiation the context's previous one ended with: scans of one sensor follow each other, and
    // the certificate's instantiation decides on the device, pass by pass, whether it tracks anything -- started from the first chunk it
    // certifies from iteration ~12 of a TUM-shaped pair instead of 17; the results are the same bits either way)
    // (the hint holds for the whole call: the certificate's instantiation costs a clean pair ~1.5 % of its passes, and the next call's
    // hint is what THIS call's finalize launches last reported)
    const bool sticky = ctx->icp_cert_hint && nchunks >= 3;          // (a call of one or two chunks never reads a word: no hint in, none out)
    ctx->icp_cert = sticky;
    struct Hint { tc_context *c; bool wanted; ~Hint() { c->icp_cert_hint = wanted; } } hint{ctx, sticky};
    for (size_t c = 0; c < nchunks; ++c) {
        if (c >= 2 && c - 2 < max_flags) {
            // wait for chunk c - 2 (chunk c - 1 keeps the device busy meanwhile); spins, then yields the core (wait_pinned_word)
            if (tc_status s = wait_pinned_word(ctx, (volatile uint32_t *)&flags[c - 2], "ICP loop")) return s;
            if (flags[c - 2] == 1) break;
            // (3: the registration wants the second-neighbour certificate -- small updates and still searching, compose() -- : the chunks
            // enqueued from here on run its instantiation of the main pass; a function of the state after chunk c - 2, not of timing)
            hint.wanted = flags[c - 2] == 3;
            ctx->icp_cert = sticky || hint.wanted;
        }
        if (c < max_flags) flags[c] = 0;
        for (size_t k = 0; k < chunk_len(c) && it < max_iters; ++k, ++it) {
            // the chunk's last launch writes the word straight into the pinned block
            const bool last = k + 1 == chunk_len(c) || it + 1 == max_iters;
            if (tc_status s = enqueue_iteration((last && c < max_flags) ? d_flags + c : nullptr)) return s;
        }
    }
    return TC_OK;
}

static tc_status icp_run_mode(tc_context *ctx, int mode, const float *d_src, size_t ns, const float *d_tgt, size_t nt,
                              const float *d_nrm, size_t nstride, const float init[7], size_t max_iters, float max_dist,
                              float conv_thr, tc_icp_result *res, bool corr_on_device, int kiss, const float *d_cov_src,
                              const float *d_cov_tgt, DeviceIndex *tgt_prebuilt, const DeviceIndex *src_presorted) {
    const bool p2plane = mode == 1;
    IcpSetup su;
    if (tc_status s = icp_setup(ctx, p2plane, d_src, ns, d_tgt, nt, d_nrm, nstride, init, max_dist, conv_thr, su, kiss, tgt_prebuilt,
                                mode == 2 ? nullptr : src_presorted)) return s;
    hipStream_t st = ctx->stream;
    IcpState *dstate = (IcpState *)ctx->state.p;
    uint32_t *corr = (uint32_t *)ctx->corr.p, *corr_pos = match_words(su.wsrc);
    double *partials = (double *)ctx->partials.p;
    const float4 *src = su.src;
    const float4 *src_cov = nullptr;
    if (mode == 2) {
        if (tc_status s = ensure(ctx, (*su.tix).normals, nt * 2 * sizeof(float4))) return s;
        if (tc_status s = ensure(ctx, ctx->gicp_src_cov, ns * 2 * sizeof(float4))) return s;
        hipLaunchKernelGGL(gather_cov_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, (const float4 *)(*su.tix).pts.p,
                           (uint32_t)nt, (const float4 *)d_cov_tgt, (float4 *)(*su.tix).normals.p);
        hipLaunchKernelGGL(gather_cov_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, src, (uint32_t)ns,
                           (const float4 *)d_cov_src, (float4 *)ctx->gicp_src_cov.p);
        src_cov = (const float4 *)ctx->gicp_src_cov.p;
    }
    const float4 *nrm = (const float4 *)(*su.tix).normals.p;

    size_t enq = 0;
    // bounds that already exist (a cloud handle whose normals were estimated here, or that has been a target before) serve from
    // the first iteration on; otherwise they are computed once the registration has run for a while
    const float4 *vor = su.tix->vor_valid ? (const float4 *)su.tix->vor.p : nullptr;
    if (debug_flags() & 1024)
        if (tc_status s = ensure(ctx, ctx->dbg_times, 10 * (size_t)kMaxPartialBlocks * sizeof(unsigned long long))) return s;
    if (tc_status s = run_chunked(ctx, max_iters, dstate, [&](int32_t *done_out) -> tc_status {
            if (enq++ == vor_after() && !vor)
                if (tc_status s = launch_target_nn_bounds(ctx, (*su.tix), su.tv, dstate, &vor)) return s;
            launch_iteration(ctx, mode, su.tv, nrm, su.wsrc, (uint32_t)ns, su.l, dstate, corr_pos, corr + 2 * ns, partials, true, true, true, src_cov, vor, done_out, su.wl, ctx->icp_cert);
            return TC_OK;
        })) return s;
    if (mode == 0) {
        ProfScope ps(ctx, "icp_final_mse");
        hipLaunchKernelGGL(icp_final_mse_kernel, dim3(su.l.mse_blocks), dim3(kIcpBlock), 0, st, su.tv, (const float4 *)su.wsrc, (uint32_t)ns,
                           su.l.mse_chunk, dstate, corr_pos, partials);
    }
    // (the RESULT block: pinned bytes 640 .. 1008 -- not the block at 256 the initial state was staged in, whose asynchronous upload
    // may still be pending when a short run has been enqueued in one go)
    IcpState *hs = (IcpState *)((char *)ctx->pinned + 640);
    IcpState *hs_dev = (IcpState *)pinned_dev_ptr(ctx, hs);
    if (!hs_dev) return fail(ctx, TC_GPU, "ICP: the pinned block has no device address");
    hs->status = TC_GPU; hs->iterations = 0;          // (overwritten by the finish kernel; a launch that never ran must not read as a result)
    hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(64), 0, st, dstate, partials, su.l.mse_blocks, mode != 0 ? 1 : 0, 0, pinned_poll_enabled() ? hs_dev : nullptr);
    if (!pinned_poll_enabled()) TC_HIP_TRY(ctx, hipMemcpyAsync(hs, dstate, sizeof(IcpState), hipMemcpyDeviceToHost, st));
    if (res->corr_target) {
        // (a caller's device array is written directly, a host array through the context's buffer)
        hipLaunchKernelGGL(icp_write_corr_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, su.tv, src, (uint32_t)ns,
                           corr_pos, corr_on_device ? res->corr_target : corr);
        if (!corr_on_device) TC_HIP_TRY(ctx, hipMemcpyAsync(res->corr_target, corr, ns * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    TC_HIP_TRY(ctx, hipGetLastError());
    if ((debug_flags() & 1024) && ctx->dbg_times.p) {          // block schedule of the LAST main pass of the call
        std::vector<unsigned long long> h(2 * (size_t)su.l.nblocks);
        (void)hipMemcpy(h.data(), ctx->dbg_times.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t b = 0; b < su.l.nblocks; ++b) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); }
        std::vector<double> dur, endt, startt;
        for (size_t b = 0; b < su.l.nblocks; ++b) { dur.push_back((h[2 * b + 1] - h[2 * b]) * 0.01); endt.push_back((h[2 * b + 1] - t0) * 0.01); startt.push_back((h[2 * b] - t0) * 0.01); }
        std::sort(dur.begin(), dur.end()); std::sort(endt.begin(), endt.end()); std::sort(startt.begin(), startt.end());
        auto q = [](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        fprintf(stderr, "[tc] main pass blocks: %u; span %.1f us; block duration us min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; start p50 %.1f p99 %.1f max %.1f; end p10 %.1f p50 %.1f p90 %.1f\n",
                su.l.nblocks, (t1 - t0) * 0.01, dur.front(), q(dur, 0.1), q(dur, 0.5), q(dur, 0.9), q(dur, 0.99), dur.back(), q(startt, 0.5), q(startt, 0.99), startt.back(),
                q(endt, 0.1), q(endt, 0.5), q(endt, 0.9));
#ifdef TC_PHASE_STAMPS
        {
            std::vector<unsigned long long> hp(8 * (size_t)su.l.nblocks);
            (void)hipMemcpy(hp.data(), (unsigned long long *)ctx->dbg_times.p + 2 * kMaxPartialBlocks, hp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double m[8] = {0};
            for (size_t b = 0; b < su.l.nblocks; ++b) for (int i = 0; i < 8; ++i) m[i] += (double)hp[8 * b + i] / su.l.nblocks;
            fprintf(stderr, "[tc]   wave 0 phases, mean s_memtime ticks: stage1-2 %.0f  windows %.0f  loop %.0f  post %.0f  stage4 %.0f  fold %.0f  end %.0f\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6]);
        }
#endif
        fprintf(stderr, "[tc]   per XCD (block & 7): mean duration / last end, us:");
        for (int x = 0; x < 8; ++x) {
            double sum = 0, last = 0; int cnt = 0;
            for (size_t b = x; b < su.l.nblocks; b += 8) { sum += (h[2 * b + 1] - h[2 * b]) * 0.01; last = std::max(last, (h[2 * b + 1] - t0) * 0.01); ++cnt; }
            fprintf(stderr, " %.1f/%.1f", sum / std::max(cnt, 1), last);
        }
        fprintf(stderr, "\n");
    }
    fold_search_stats(ctx, hs);
    if (debug_flags() & 64)
        fprintf(stderr, "[tc] icp: searching lanes at the last evaluation of the certificate's gate %u (d_run %u, wanted %d)\n", hs->searchers, hs->d_run, hs->d_ang >= 0.0f ? 1 : 0);
    if (debug_flags() & 64)
        fprintf(stderr, "[tc] icp: %u iterations, refine queries total %u max %u  exit ring hist %u %u %u %u %u %u %u %u\n", hs->iterations,
                hs->refine_total, hs->refine_max, hs->refine_ring_hist[0], hs->refine_ring_hist[1], hs->refine_ring_hist[2], hs->refine_ring_hist[3],
                hs->refine_ring_hist[4], hs->refine_ring_hist[5], hs->refine_ring_hist[6], hs->refine_ring_hist[7]);
    if (hs->status != TC_OK) {
        return fail(ctx, (tc_status)hs->status,
                    mode == 2 ? "GICP: insufficient correspondences (need >= 6) or ill-conditioned Gauss-Newton system"
                    : p2plane ? "Insufficient correspondences for point-to-plane ICP (need >= 6) or ill-conditioned system"
                            : "Insufficient correspondences found");
    }
    if (tc_status s = check_iteration_count(ctx, hs, max_iters)) return s;
    for (int i = 0; i < 4; ++i) res->transformation[i] = hs->q[i];
    for (int i = 0; i < 3; ++i) res->transformation[4 + i] = hs->t[i];
    res->mse = hs->mse;
    res->iterations = hs->converged ? hs->iterations : max_iters;
    res->converged = hs->converged;
    res->n_correspondences = hs->n_corr;
    return TC_OK;
}


// ---- one registration over the ranks of a communicator -----------------------------------------------
// Same kernels as the single-GPU loop; per iteration
//     main + refine (this rank's shard)  ->  icp_finalize(sum only): 29 / 17 words in IcpState::sums
//     ->  all-reduce(sum) of TC_ICP_SUMS_STRIDE doubles in place, on the context's stream (RCCL)
//     ->  icp_finalize(apply only): every rank solves the identical system -> identical state, no broadcast.
// nranks == 1: the all-reduce is skipped and the sums, the solve and therefore every bit of the result equal the fused
// loop's (the finalize kernel folds the same rows in the same order whether or not it also applies them).
__global__ void __launch_bounds__(256) icp_zero_u32_kernel(uint32_t *__restrict__ p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

tc_status icp_run_sharded(tc_context *ctx, tc_comm *comm, int shard_mode, bool p2plane, const float *d_src, size_t ns, const float *d_tgt,
                          size_t nt, const float *d_nrm, size_t nstride, const float init[7], size_t max_iters, float max_dist,
                          float conv_thr, tc_icp_result *res, DeviceIndex *tgt_prebuilt) {
    const int mode = p2plane ? 1 : 0;
    const int W = comm ? comm->nranks : 1, rank = comm ? comm->rank : 0;
    IcpSetup su;
    // TC_SHARD_INDEX: this rank's ORIGINAL-index range of the replicated source -- only those points are ordered by target cell
    // (the per-call set-up shrinks with 1 / W; TC_SHARD_SPATIAL orders the whole source on every rank and takes a range of the order)
    const size_t rows = (ns + (size_t)W - 1) / (size_t)W;
    size_t ilo = 0, ihi = ns;
    if (shard_mode == TC_SHARD_INDEX) { ilo = std::min((size_t)rank * rows, ns); ihi = std::min(ilo + rows, ns); }
    const size_t n_setup = ihi - ilo;              // source points this rank's set-up sees
    // (TC_SHARD_SPATIAL over several ranks: every rank sorts the replicated source and takes a range of positions -- the ranks must
    // agree on the order, also inside a cell too populous for the deterministic re-rank: build_index(strict_order))
    tc_status setup_rc = icp_setup(ctx, p2plane, d_src + 3 * ilo, n_setup, d_tgt, nt, d_nrm, nstride, init, max_dist, conv_thr, su, 0, tgt_prebuilt, nullptr,
                                   shard_mode == TC_SHARD_SPATIAL && W > 1);
    // Everything that can fail on ONE rank happens before the first collective, and the ranks agree on it: the inscribed-ball
    // bounds' buffer (the loop computes them at iteration 6) and the gathered correspondence array are allocated here, so that no
    // fallible host path sits between two collectives -- a rank returning early from inside the loop would leave its peers waiting
    // in ncclAllReduce for ever.
    if (setup_rc == TC_OK && !su.tix->vor_valid) setup_rc = ensure(ctx, su.tix->vor, (size_t)su.tix->geom.n * sizeof(float4));
    const bool gather_index = shard_mode == TC_SHARD_INDEX && res->corr_target && ns > 0 && (W > 1 || (comm && comm->nccl));
    if (setup_rc == TC_OK && gather_index) setup_rc = ensure(ctx, ctx->out_a, (size_t)W * rows * sizeof(uint32_t));
    if (tc_status s = comm_agree(comm, setup_rc)) return s;
    hipStream_t st = ctx->stream;
    IcpState *dstate = (IcpState *)ctx->state.p;
    // this rank's range of the tile-major sorted source: a spatially compact shard (TC_SHARD_SPATIAL), or everything it set up
    size_t lo = 0, hi = n_setup;
    if (shard_mode == TC_SHARD_SPATIAL) { lo = ns * (size_t)rank / (size_t)W; hi = ns * ((size_t)rank + 1) / (size_t)W; }
    const size_t nl = hi - lo;
    const IcpLaunch l = plan_launch(nl);
    // corr (n_setup) | corr_pos (nl) | refine counts + entries -- inside the buffer icp_setup sized for the points it was given
    uint32_t *corr = (uint32_t *)ctx->corr.p, *rlist = corr + 2 * n_setup;
    double *partials = (double *)ctx->partials.p;
    const float4 *src = su.src + lo;
    float4 *wsrc = su.wsrc + lo;
    uint32_t *corr_pos = match_words(wsrc);
    const float4 *nrm = (const float4 *)(*su.tix).normals.p;
    size_t enq = 0;
    // bounds that already exist (a cloud handle whose normals were estimated here, or that has been a target before) serve from
    // the first iteration on; otherwise they are computed once the registration has run for a while
    const float4 *vor = su.tix->vor_valid ? (const float4 *)su.tix->vor.p : nullptr;
    if (tc_status s = run_chunked(ctx, max_iters, dstate, [&](int32_t *done_out) -> tc_status {
            if (enq++ == vor_after() && !vor)
                if (tc_status s = launch_target_nn_bounds(ctx, (*su.tix), su.tv, dstate, &vor)) return s;
            launch_iteration(ctx, mode, su.tv, nrm, wsrc, (uint32_t)nl, l, dstate, corr_pos, rlist, partials, true, false, true, nullptr, vor);
            if (tc_status s = comm_allreduce_f64(comm, dstate->sums, TC_ICP_SUMS_STRIDE)) return s;
            launch_iteration(ctx, mode, su.tv, nrm, wsrc, (uint32_t)nl, l, dstate, corr_pos, rlist, partials, false, true, false, nullptr, nullptr, done_out);
            return TC_OK;
        })) return s;
    if (mode == 0) {        // registration.rs:343-361: the post-loop mse of a run that did not converge, summed over the ranks
        hipLaunchKernelGGL(icp_final_mse_kernel, dim3(l.mse_blocks), dim3(kIcpBlock), 0, st, su.tv, (const float4 *)wsrc, (uint32_t)nl, l.mse_chunk, dstate,
                           corr_pos, partials);
        hipLaunchKernelGGL(icp_final_mse_fold_kernel, dim3(1), dim3(64), 0, st, dstate, partials, l.mse_blocks);
        if (tc_status s = comm_allreduce_f64(comm, dstate->sums, TC_ICP_SUMS_STRIDE)) return s;
    }
    IcpState *hs = (IcpState *)((char *)ctx->pinned + 640);        // (the result block, see icp_run_mode)
    IcpState *hs_dev = (IcpState *)pinned_dev_ptr(ctx, hs);
    if (!hs_dev) return fail(ctx, TC_GPU, "ICP: the pinned block has no device address");
    hs->status = TC_GPU; hs->iterations = 0;
    hipLaunchKernelGGL(icp_finish_kernel, dim3(1), dim3(64), 0, st, dstate, partials, 0u, mode != 0 ? 1 : 0, 1, pinned_poll_enabled() ? hs_dev : nullptr);
    if (!pinned_poll_enabled()) TC_HIP_TRY(ctx, hipMemcpyAsync(hs, dstate, sizeof(IcpState), hipMemcpyDeviceToHost, st));
    if (res->corr_target && ns > 0) {
        if (gather_index) {
            // every rank writes the matches of ITS index range (the write-out kernel scatters by the index inside the slice) into its
            // slot of `rows` entries; one in-place all-gather of the slots completes the dense array everywhere: 4 ns bytes in total
            // (the spatial mode's all-reduce moves ns words per rank through a reduction)
            uint32_t *all = (uint32_t *)ctx->out_a.p;
            if (nl > 0)
                hipLaunchKernelGGL(icp_write_corr_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, su.tv, src, (uint32_t)nl, corr_pos,
                                   all + (size_t)rank * rows);
            if (tc_status s = comm_allgather(comm, all, rows * sizeof(uint32_t))) return s;
            TC_HIP_TRY(ctx, hipMemcpyAsync(res->corr_target, all, ns * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        } else {
            const bool gather = shard_mode == TC_SHARD_SPATIAL && (W > 1 || (comm && comm->nccl));
            // the write-out kernel scatters by ORIGINAL source index: with the source sharded spatially every rank fills its own
            // entries of a zeroed array and one all-reduce(sum) of n_source words completes it everywhere
            if (gather) hipLaunchKernelGGL(icp_zero_u32_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, corr, (uint32_t)ns);
            if (nl > 0)
                hipLaunchKernelGGL(icp_write_corr_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, su.tv, src, (uint32_t)nl, corr_pos, corr);
            if (gather)
                if (tc_status s = comm_allreduce_u32(comm, corr, ns)) return s;
            TC_HIP_TRY(ctx, hipMemcpyAsync(res->corr_target, corr, n_setup * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        }
    }
    TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    TC_HIP_TRY(ctx, hipGetLastError());
    fold_search_stats(ctx, hs);
    if (hs->status != TC_OK)
        return fail(ctx, (tc_status)hs->status, p2plane ? "Insufficient correspondences for point-to-plane ICP (need >= 6) or ill-conditioned system"
                                                        : "Insufficient correspondences found");
    if (tc_status s = check_iteration_count(ctx, hs, max_iters)) return s;
    for (int i = 0; i < 4; ++i) res->transformation[i] = hs->q[i];
    for (int i = 0; i < 3; ++i) res->transformation[4 + i] = hs->t[i];
    res->mse = hs->mse;
    res->iterations = hs->converged ? hs->iterations : max_iters;
    res->converged = hs->converged;
    res->n_correspondences = hs->n_corr;
    return TC_OK;
}

}  // namespace tc

// ---- sharded building blocks (C ABI) ---------------------------------------------------------
struct tc_icp_shard {
    tc_context *ctx;
    bool p2plane;
    size_t ns;
    tc::IcpSetup su;
};

extern "C" {

tc_status tc_icp_shard_create(tc_context *ctx, int point_to_plane, const float *d_source_slice, size_t n_source_slice,
                              const float *d_target, size_t n_target, const float *d_target_normals, size_t normal_stride,
                              const float init[7], float max_correspondence_distance, float convergence_threshold,
                              tc_icp_shard **out) try {
    if (!ctx || !out) return TC_INVALID_DATA;
    if (n_source_slice == 0 || n_target == 0) return tc::fail(ctx, TC_INVALID_DATA, "Source or target point cloud is empty");
    // the step-wise building blocks have no cross-rank post-loop mse recompute (registration.rs:343-361):
    // point-to-point runs go through tc_sharded_icp_detailed_device
    if (!point_to_plane) return tc::fail(ctx, TC_UNSUPPORTED, "tc_icp_shard_*: point-to-plane only; use tc_sharded_icp_detailed_device");
    tc_icp_shard *s = new tc_icp_shard{ctx, point_to_plane != 0, n_source_slice, {}};
    tc_status rc = tc::icp_setup(ctx, s->p2plane, d_source_slice, n_source_slice, d_target, n_target, d_target_normals,
                                 normal_stride, init, max_correspondence_distance, convergence_threshold, s->su);
    if (rc != TC_OK) { delete s; return rc; }
    *out = s;
    return TC_OK;
} TC_CATCH_STATUS(ctx)

double *tc_icp_shard_sums(tc_icp_shard *s) { return ((tc::IcpState *)s->ctx->state.p)->sums; }

tc_status tc_icp_shard_get_sums(tc_icp_shard *s, double *d_out) try {
    tc_context *ctx = s->ctx;
    TC_HIP_TRY(ctx, hipMemcpyAsync(d_out, ((tc::IcpState *)ctx->state.p)->sums, TC_ICP_SUMS_STRIDE * sizeof(double),
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_icp_shard_set_sums(tc_icp_shard *s, const double *d_in) try {
    tc_context *ctx = s->ctx;
    TC_HIP_TRY(ctx, hipMemcpyAsync(((tc::IcpState *)ctx->state.p)->sums, d_in, TC_ICP_SUMS_STRIDE * sizeof(double),
                                   hipMemcpyDeviceToDevice, ctx->stream));
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_icp_shard_done(tc_icp_shard *s, int *done) try {
    tc_context *ctx = s->ctx;
    int32_t *h = (int32_t *)((char *)ctx->pinned + 1024);
    TC_HIP_TRY(ctx, hipMemcpyAsync(h, &((tc::IcpState *)ctx->state.p)->done, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    TC_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *done = *h;
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_icp_shard_reduce(tc_icp_shard *s) try {
    tc_context *ctx = s->ctx;
    uint32_t *corr = (uint32_t *)ctx->corr.p;
    tc::launch_iteration(ctx, s->p2plane, s->su.tv, (const float4 *)ctx->tgt_index.normals.p, s->su.wsrc,
                         (uint32_t)s->ns, s->su.l, (tc::IcpState *)ctx->state.p, tc::match_words(s->su.wsrc), corr + 2 * s->ns,
                         (double *)ctx->partials.p, true, false, true);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_icp_shard_apply(tc_icp_shard *s) try {
    tc_context *ctx = s->ctx;
    uint32_t *corr = (uint32_t *)ctx->corr.p;
    tc::launch_iteration(ctx, s->p2plane, s->su.tv, (const float4 *)ctx->tgt_index.normals.p, s->su.wsrc,
                         (uint32_t)s->ns, s->su.l, (tc::IcpState *)ctx->state.p, tc::match_words(s->su.wsrc), corr + 2 * s->ns,
                         (double *)ctx->partials.p, false, true, false);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

tc_status tc_icp_shard_finish(tc_icp_shard *s, size_t max_iters, tc_icp_result *res) try {
    tc_context *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    tc::IcpState *dstate = (tc::IcpState *)ctx->state.p;
    // point-to-point's post-loop mse recompute needs a cross-rank sum: the host driver does it
    // (threecrate_amd.distributed); here the p2plane rule (mse = previous_mse) is applied.
    hipLaunchKernelGGL(tc::icp_finish_kernel, dim3(1), dim3(64), 0, st, dstate, (const double *)ctx->partials.p, 0u, 1, 0, (tc::IcpState *)nullptr);
    tc::IcpState *hs = (tc::IcpState *)((char *)ctx->pinned + 256);
    TC_HIP_TRY(ctx, hipMemcpyAsync(hs, dstate, sizeof(tc::IcpState), hipMemcpyDeviceToHost, st));
    if (res->corr_target) {
        uint32_t *corr = (uint32_t *)ctx->corr.p;
        hipLaunchKernelGGL(tc::icp_write_corr_kernel, dim3((unsigned)((s->ns + 255) / 256)), dim3(256), 0, st, s->su.tv,
                           (const float4 *)ctx->src_index.pts.p, (uint32_t)s->ns, tc::match_words(s->su.wsrc), corr);
        TC_HIP_TRY(ctx, hipMemcpyAsync(res->corr_target, corr, s->ns * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    }
    TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    tc::fold_search_stats(ctx, hs);
    if (hs->status != TC_OK) return tc::fail(ctx, (tc_status)hs->status, "ICP failed (insufficient correspondences / singular system)");
    for (int i = 0; i < 4; ++i) res->transformation[i] = hs->q[i];
    for (int i = 0; i < 3; ++i) res->transformation[4 + i] = hs->t[i];
    res->mse = hs->mse;
    res->iterations = hs->converged ? hs->iterations : max_iters;
    res->converged = hs->converged;
    res->n_correspondences = hs->n_corr;
    return TC_OK;
} TC_CATCH_STATUS((s ? s->ctx : nullptr))

void tc_icp_shard_destroy(tc_icp_shard *s) { delete s; }

}  // extern "C"
