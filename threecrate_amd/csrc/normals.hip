// normals.hip -- HK2 normals_knn_pca: fused exact grid k-NN -> covariance -> 3x3 eigenvector
// -> viewpoint orientation.  One lane per point, points visited in cell-sorted order.
//
// Reference semantics (threecrate-algorithms/src/normals.rs):
//   neighbours = kNN(k+1) minus self, first k, ascending distance (:147-153); self appended
//   last (:338-340); f32 centroid and covariance in that order (:164-177); eigenvector of the
//   smallest eigenvalue (:181-194); renormalise (:197-202); flip towards viewpoint (:208-222).
//
// Device algorithm per point (no neighbour list ever reaches HBM):
//   phase 1  scan the (2R+1)^3 cell block as (2R+1)^2 contiguous spans of the cell-sorted
//            record array; keep the L smallest squared distances in a sorted REGISTER list.
//            Insertion into the sorted list is branch-free: new[t] = med3(old[t-1], v, old[t]),
//            one v_med3_f32 per slot, all slots independent.
//   exactness tau = (k+1)-th smallest d2 is final iff tau <= (R*h + m)^2, m = distance from
//            the query to the nearest face of its own cell (every unscanned point is at least
//            R*h + m away).  Lanes that fail (grid-boundary points, Poisson tail: ~1 %) continue
//            in place with the next shell, ball-pruned: only cells within sqrt(current tau) of the
//            query are visited, and a shell that misses that ball ends the search.
//   phase 2  rescan the cells within sqrt(tau), append the record positions with d2 <= tau to a per-lane LDS list
//            (ties at tau: lowest cell-sorted position first), rank them against the register
//            list -> neighbours in ascending-distance order, exactly the reference's order.
//   epilogue f32 centroid / covariance in the reference's operation order, eigenvectors by nalgebra's f32
//            symmetric_eigen algorithm (sym_eigen3_f32), column of the smallest eigenvalue; the radius-set path keeps an
//            f64 closed form (trigonometric estimate + monotone Newton polish + largest cross product).
//
//   round 3  on volumetric indexes the list carries the neighbours' positions in the low bits of its keys and the rows of the
//            block are walked as three flattened groups (knn_tagged below: no phase 2; the register-list path above is its
//            fallback, and the path for surfaces, dense cells, radius mode and lists beyond 23 entries).
//
// Algorithmic HBM bytes per point: 12 (query) + 12*k (neighbours) + 24 (out).
#include "tc_internal.h"

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace tc {

struct NormalParams {
    uint32_t k;             // k_neighbors
    int      orient;        // consistent_orientation
    float    vx, vy, vz;    // viewpoint
    int      R0;            // ring count of the main launch
    int      has_radius;
    float    radius;
    const float *xyz;       // the caller's AoS input (positions of non-finite points are copied from here)
    float4 *sorted_nrm;     // optional: the normal of cell-sorted position p -> sorted_nrm[p] (a cloud handle keeps them: the ICP target layout)
    float4 *vor_out;        // optional: {x, y, z, inscribed-ball bound} per position (icp.hip: the bound is a quarter of the squared
                            // distance to the nearest OTHER record = the second entry of the k-NN list, for free here)
    uint32_t *hard_list;        // optional: positions whose search would outgrow kHardRing rings are appended here (hard_list[0] = count, entries from [4];
                                // [1] = the serving kernel's exit ticket) and served by normals_coop_kernel, a block per point
    uint32_t p_begin, p_end;    // cell-sorted positions handled by this launch (a multi-GPU shard: SURVEY 8e)
    int      slice_out;         // 1: record of position p goes to row p - p_begin (sorted order) instead of its original index
    int      tag_policy;        // tagged-key kernels: 1 = take the tagged path only on a volumetric index (decided on the device from the
                                // occupied-cell count build_index leaves in front of the prefix sums), 0 = always try it
#ifdef TC_PHASE_STAMPS
    unsigned long long *stamps; // dev build: 8 per block (wave 0's shader clocks per phase)
#endif
};
#ifdef TC_PHASE_STAMPS
#define TC_NSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tl; tl = t_; } while (0)
#else
#define TC_NSTAMP(i) do { } while (0)
#endif

// ---- smallest-eigenvalue eigenvector of a symmetric 3x3 (f64) -------------------------------
__device__ __forceinline__ void smallest_eigvec_sym3(double a00, double a01, double a02, double a11, double a12,
                                                     double a22, double &nx, double &ny, double &nz) {
    double scale = fmax(fmax(fmax(fabs(a00), fabs(a01)), fmax(fabs(a02), fabs(a11))), fmax(fabs(a12), fabs(a22)));
    if (!(scale > 0.0)) { nx = 1.0; ny = 0.0; nz = 0.0; return; }   // zero matrix: reference's Q = I, column 0
    double inv = 1.0 / scale;
    a00 *= inv; a01 *= inv; a02 *= inv; a11 *= inv; a12 *= inv; a22 *= inv;
    double q = (a00 + a11 + a22) * (1.0 / 3.0);
    double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
    double p = sqrt(p2 * (1.0 / 6.0));
    if (!(p > 1e-150)) { nx = 1.0; ny = 0.0; nz = 0.0; return; }    // multiple of the identity
    // f32 trigonometric estimate of the smallest root
    double ip = 1.0 / p;
    double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    double detB = c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02);
    float r = (float)(0.5 * detB);
    r = fminf(1.0f, fmaxf(-1.0f, r));
    float phi = acosf(r) * (1.0f / 3.0f);
    double lam = q + 2.0 * p * (double)cosf(phi + 2.0943951023931953f);
    // characteristic polynomial det(A - x I) = -x^3 + c2 x^2 - c1 x + c0
    double c2 = a00 + a11 + a22;
    double c1 = (a00 * a11 - a01 * a01) + (a00 * a22 - a02 * a02) + (a11 * a22 - a12 * a12);
    double c0 = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    // start strictly below the smallest root: det(A - xI) is positive, decreasing and convex
    // there, so Newton converges monotonically from the left.
    lam -= 4e-6 * (p + fabs(q));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        double f = ((-lam + c2) * lam - c1) * lam + c0;
        double df = (-3.0 * lam + 2.0 * c2) * lam - c1;
        if (df < 0.0) lam -= f / df;
    }
    double m00 = a00 - lam, m11 = a11 - lam, m22 = a22 - lam;
    // cross products of the rows of (A - lam I)
    double x0 = a01 * a12 - a02 * m11, y0 = a02 * a01 - m00 * a12, z0 = m00 * m11 - a01 * a01;      // r0 x r1
    double x1 = a01 * m22 - a02 * a12, y1 = a02 * a02 - m00 * m22, z1 = m00 * a12 - a01 * a02;      // r0 x r2
    double x2 = m11 * m22 - a12 * a12, y2 = a12 * a02 - a01 * m22, z2 = a01 * a12 - m11 * a02;      // r1 x r2
    double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1, n2 = x2 * x2 + y2 * y2 + z2 * z2;
    double bx = x0, by = y0, bz = z0, bn = n0;
    if (n1 > bn) { bx = x1; by = y1; bz = z1; bn = n1; }
    if (n2 > bn) { bx = x2; by = y2; bz = z2; bn = n2; }
    if (bn > 1e-28) {
        double s = 1.0 / sqrt(bn);
        nx = bx * s; ny = by * s; nz = bz * s;
        return;
    }
    // rank <= 1 (two smallest eigenvalues coincide): any unit vector orthogonal to the dominant row
    double r0n = m00 * m00 + a01 * a01 + a02 * a02, r1n = a01 * a01 + m11 * m11 + a12 * a12,
           r2n = a02 * a02 + a12 * a12 + m22 * m22;
    double ux = m00, uy = a01, uz = a02, un = r0n;
    if (r1n > un) { ux = a01; uy = m11; uz = a12; un = r1n; }
    if (r2n > un) { ux = a02; uy = a12; uz = m22; un = r2n; }
    if (!(un > 0.0)) { nx = 1.0; ny = 0.0; nz = 0.0; return; }
    // pick the coordinate axis least aligned with u and orthogonalise
    double ax = fabs(ux), ay = fabs(uy), az = fabs(uz);
    double ex = (ax <= ay && ax <= az) ? 1.0 : 0.0, ey = (ex == 0.0 && ay <= az) ? 1.0 : 0.0;
    double ez = (ex == 0.0 && ey == 0.0) ? 1.0 : 0.0;
    double vx = uy * ez - uz * ey, vy = uz * ex - ux * ez, vz = ux * ey - uy * ex;
    double s = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
    nx = vx * s; ny = vy * s; nz = vz * s;
}

// ---- nalgebra's Matrix3::symmetric_eigen in f32, operation by operation (normals.rs:181) -----------------------
// Scale by the max-abs entry, Householder tridiagonalisation, implicit symmetric QR with Wilkinson shifts, direct 2x2
// solve for the last block; eigenvalues unsorted, eigenvectors = columns of q.  The eigenvector of a near-degenerate
// neighbourhood (two smallest eigenvalues close) is whatever THIS arithmetic produces -- a more accurate solver gives a
// different, equally valid vector, i.e. no parity on those points -- so the k-NN path runs the reference's algorithm in
// the reference's precision (nalgebra 0.34 linalg: SymmetricTridiagonal::new, householder::assemble_q, SymmetricEigen::
// do_decompose / delimit_subproblem, GivensRotation::cancel_y / try_new, wilkinson_shift -- same order of operations; the library is
// built without FMA contraction).  Only the lower triangle of the input is read.
__device__ __forceinline__ void e3_to_exp(float x, float &mod, float &sign) {
    const float n = fabsf(x);
    if (n != 0.0f) { mod = n; sign = x / n; } else { mod = 0.0f; sign = 1.0f; }
}
__device__ __forceinline__ bool e3_givens_cancel_y(float x, float y, float &c, float &s, float &r) {
    if (y != 0.0f) {
        float mod0, sign0;
        e3_to_exp(x, mod0, sign0);
        const float denom = sqrtf(mod0 * mod0 + y * y);
        c = mod0 / denom;
        s = -y / (sign0 * denom);
        r = sign0 * denom;
        return true;
    }
    return false;
}
__device__ __forceinline__ float e3_wilkinson_shift(float tmm, float tnn, float tmn) {
    const float sq = tmn * tmn;
    if (sq != 0.0f) {
        const float d = (tmm - tnn) * 0.5f;
        float sg = (d >= 0.0f || d != d) ? 1.0f : -1.0f;
        if (d == 0.0f && signbit(d)) sg = -1.0f;
        return tnn - sq / (d + sg * sqrtf(d * d + sq));
    }
    return tnn;
}
// SymmetricEigen::delimit_subproblem for n = 3 (indices spelled out: no dynamic register indexing)
__device__ __forceinline__ void e3_delimit(float d0, float d1, float d2, float &o0, float &o1, int end, float eps, int &start_out, int &end_out) {
    int n = end;
    if (n == 2 && !(fabsf(o1) > eps * (fabsf(d2) + fabsf(d1)))) n = 1;
    if (n == 1 && !(fabsf(o0) > eps * (fabsf(d1) + fabsf(d0)))) n = 0;
    if (n == 0) { start_out = 0; end_out = 0; return; }
    int ns = n - 1;
    if (ns == 1) {
        if (o0 == 0.0f || fabsf(o0) <= eps * (fabsf(d1) + fabsf(d0))) o0 = 0.0f;
        else ns = 0;
    }
    start_out = ns; end_out = n;
}
// q <- q * [[c, s], [-s, c]] on two columns (scalars, not an array: the final "column of the smallest eigenvalue" select
// over an array turns into a dynamically indexed scratch load)
__device__ __forceinline__ void e3_rot_cols(float &a0, float &a1, float &a2, float &b0, float &b1, float &b2, float c, float s) {
    float qa = a0, qb = b0; a0 = qa * c - s * qb; b0 = s * qa + qb * c;
    qa = a1; qb = b1; a1 = qa * c - s * qb; b1 = s * qa + qb * c;
    qa = a2; qb = b2; a2 = qa * c - s * qb; b2 = s * qa + qb * c;
}
__device__ __forceinline__ void e3_rot_cols_t(float &a0, float &a1, float &a2, float &b0, float &b1, float &b2, float c, float s) {   // the final 2x2 block's rotation
    float qa = a0, qb = b0; a0 = qa * c + s * qb; b0 = -s * qa + qb * c;
    qa = a1; qb = b1; a1 = qa * c + s * qb; b1 = -s * qa + qb * c;
    qa = a2; qb = b2; a2 = qa * c + s * qb; b2 = -s * qa + qb * c;
}
// one implicit-QR Givens step on (d_i, d_j, o_i)
__device__ __forceinline__ void e3_qr_update(float &di, float &dj, float &oi, float c, float s) {
    const float mii = di, mjj = dj, mij = oi;
    const float cc = c * c, ss = s * s, cs = c * s;
    const float b = cs * 2.0f * mij;
    di = (cc * mii + ss * mjj) - b;
    dj = (ss * mii + cc * mjj) + b;
    oi = cs * (mii - mjj) + mij * (cc - ss);
}
__device__ __forceinline__ void e3_last_block(float &ds, float &ds1, float os, float eps, float &c, float &s, bool &rotate) {
    const float h00 = ds, h10 = os, h11 = ds1;
    const float val = (h00 - h11) * 0.5f;
    const float discr = h10 * h10 + val * val;
    const float sq = sqrtf(discr);
    const float half_tra = (h00 + h11) * 0.5f;
    const float e0 = half_tra + sq, e1 = half_tra - sq;
    const float bx = e0 - ds1, by = os;
    ds = e0; ds1 = e1;
    float mod0, sign0;
    e3_to_exp(bx, mod0, sign0);
    const float denom = sqrtf(mod0 * mod0 + by * by);
    rotate = denom > eps;
    if (rotate) { c = mod0 / denom; s = by / (sign0 * denom); }
}

// out: the three eigenvalues (unsorted) and q column by column: (x0, y0, z0) belongs to e0, ...
__device__ __forceinline__ void sym_eigen3_f32(float axx, float axy, float axz, float ayy, float ayz, float azz, float &e0, float &e1,
                                               float &e2, float &q00, float &q10, float &q20, float &q01, float &q11, float &q21,
                                               float &q02, float &q12, float &q22) {
    // a[i][j] (lower triangle): a00 = xx, a10 = xy, a20 = xz, a11 = yy, a21 = yz, a22 = zz; the max runs over all nine entries
    float amax = fmaxf(fmaxf(fmaxf(fabsf(axx), fabsf(axy)), fmaxf(fabsf(axz), fabsf(ayy))), fmaxf(fabsf(ayz), fabsf(azz)));
    float a00 = axx, a10 = axy, a20 = axz, a11 = ayy, a21 = ayz, a22 = azz;
    if (amax != 0.0f) { a00 = a00 / amax; a10 = a10 / amax; a20 = a20 / amax; a11 = a11 / amax; a21 = a21 / amax; a22 = a22 / amax; }
    float offs0, offs1, u0 = 0.0f, u1 = 0.0f;
    bool refl0 = false, refl1 = false;
    {   // step 0: reflect (a10, a20) onto e1
        const float x0 = a10, x1 = a20;
        const float sq = x0 * x0 + x1 * x1;
        const float nrm = sqrtf(sq);
        float mod, sign;
        e3_to_exp(x0, mod, sign);
        const float signed_norm = sign * nrm;
        const float factor = (sq + mod * nrm) * 2.0f;
        if (factor != 0.0f) {
            const float f = sqrtf(factor);
            u0 = (x0 + signed_norm) / f; u1 = x1 / f;
            const float un = sqrtf(u0 * u0 + u1 * u1);
            u0 /= un; u1 /= un;
            offs0 = -signed_norm; refl0 = true;
            float b00 = a11, b10 = a21, b11 = a22;
            const float p0 = 2.0f * (b00 * u0 + b10 * u1);
            const float p1 = 2.0f * (b10 * u0 + b11 * u1);
            const float dot = u0 * p0 + u1 * p1;
            b00 = b00 - p0 * u0; b10 = b10 - p1 * u0; b11 = b11 - p1 * u1;
            b00 = b00 - u0 * p0; b10 = b10 - u1 * p0; b11 = b11 - u1 * p1;
            const float d2 = dot * 2.0f;
            b00 = b00 + d2 * u0 * u0; b10 = b10 + d2 * u1 * u0; b11 = b11 + d2 * u1 * u1;
            a11 = b00; a21 = b10; a22 = b11;
        } else {
            offs0 = signed_norm;
        }
    }
    float ax1 = 0.0f;
    {   // step 1: the 1-vector (a21)
        const float x0 = a21;
        const float sq = x0 * x0, nrm = sqrtf(sq);
        float mod, sign;
        e3_to_exp(x0, mod, sign);
        const float signed_norm = sign * nrm;
        const float factor = (sq + mod * nrm) * 2.0f;
        if (factor != 0.0f) {
            const float f = sqrtf(factor);
            ax1 = (x0 + signed_norm) / f;
            ax1 = ax1 / fabsf(ax1);
            offs1 = -signed_norm; refl1 = true;
            float b = a22;
            const float pp = 2.0f * (b * ax1);
            const float dot = ax1 * pp;
            b = b - pp * ax1; b = b - ax1 * pp; b = b + (dot * 2.0f) * ax1 * ax1;
            a22 = b;
        } else {
            offs1 = signed_norm;
        }
    }
    float d0 = a00, d1 = a11, d2 = a22;
    float o0 = fabsf(offs0), o1 = fabsf(offs1);
    // householder::assemble_q with signs = the off-diagonal before the modulus
    // (q_rc: row r, column c)
    q00 = 1.0f; q01 = 0.0f; q02 = 0.0f; q10 = 0.0f; q11 = 1.0f; q12 = 0.0f; q20 = 0.0f; q21 = 0.0f; q22 = 1.0f;
    {
        const float sg = (offs1 < 0.0f || (offs1 == 0.0f && signbit(offs1))) ? -1.0f : 1.0f;
        const float axis = refl1 ? ax1 : 0.0f;
        auto reflect1 = [&](float &x) {                 // row 2, columns 1 and 2
            const float col = x;
            const float factor = (axis * col) * -2.0f;
            x = sg * col + axis * (factor * sg);
        };
        reflect1(q21); reflect1(q22);
    }
    {
        const float sg = (offs0 < 0.0f || (offs0 == 0.0f && signbit(offs0))) ? -1.0f : 1.0f;
        const float b0 = refl0 ? u0 : 0.0f, b1 = refl0 ? u1 : 0.0f;
        auto reflect0 = [&](float &r1, float &r2) {     // rows 1 and 2 of one column
            const float c0 = r1, c1 = r2;
            const float factor = (b0 * c0 + b1 * c1) * -2.0f;
            r1 = sg * c0 + b0 * (factor * sg);
            r2 = sg * c1 + b1 * (factor * sg);
        };
        reflect0(q10, q20); reflect0(q11, q21); reflect0(q12, q22);
    }
    // implicit QR iterations (SymmetricEigen::do_decompose)
    const float eps = 1.1920929e-07f;
    int start, end;
    e3_delimit(d0, d1, d2, o0, o1, 2, eps, start, end);
    for (int guard = 0; end != start && guard < 10000; ++guard) {
        if (end - start == 2) {                 // the full 3 x 3 problem: start = 0, end = 2
            float vx = d0 - e3_wilkinson_shift(d1, d2, o1);
            float vy = o0;
            float c, s, nrm;
            if (e3_givens_cancel_y(vx, vy, c, s, nrm)) {
                e3_qr_update(d0, d1, o0, c, s);
                vx = o0;
                vy = -s * o1;
                o1 *= c;
                e3_rot_cols(q00, q10, q20, q01, q11, q21, c, s);
                if (e3_givens_cancel_y(vx, vy, c, s, nrm)) {
                    o0 = nrm;
                    e3_qr_update(d1, d2, o1, c, s);
                    e3_rot_cols(q01, q11, q21, q02, q12, q22, c, s);
                }
            }
            if (fabsf(o1) <= eps * (fabsf(d1) + fabsf(d2))) end -= 1;
        } else {                                // a 2 x 2 block: (start, start + 1) -- on selected VALUES (two branches
            const bool lo = start == 0;         // calling one helper by reference become a dynamically indexed scratch array)
            float ds = lo ? d0 : d1, ds1 = lo ? d1 : d2;
            const float os = lo ? o0 : o1;
            float c = 1.0f, s = 0.0f;
            bool rotate;
            e3_last_block(ds, ds1, os, eps, c, s, rotate);
            d0 = lo ? ds : d0; d1 = lo ? ds1 : ds; d2 = lo ? d2 : ds1;
            if (rotate) {
                float a0 = lo ? q00 : q01, a1 = lo ? q10 : q11, a2 = lo ? q20 : q21;
                float b0 = lo ? q01 : q02, b1 = lo ? q11 : q12, b2 = lo ? q21 : q22;
                e3_rot_cols_t(a0, a1, a2, b0, b1, b2, c, s);
                q00 = lo ? a0 : q00; q10 = lo ? a1 : q10; q20 = lo ? a2 : q20;
                q01 = lo ? b0 : a0;  q11 = lo ? b1 : a1;  q21 = lo ? b2 : a2;
                q02 = lo ? q02 : b0; q12 = lo ? q12 : b1; q22 = lo ? q22 : b2;
            }
            end -= 1;
        }
        e3_delimit(d0, d1, d2, o0, o1, end, eps, start, end);
    }
    e0 = d0 * amax; e1 = d1 * amax; e2 = d2 * amax;
}

constexpr int kHardRing = 6;        // a search that would go beyond this many rings is a hard point (normals_coop_kernel)

// square root for pruning radii: the raw v_sqrt_f32 (1 ulp; the callers add the cell-assignment fuzz, thousands of ulps, as slack;
// sqrtf's correctly rounded sequence costs ~10 instructions per row: 434 -> 421 us at 1 M points)
#define TC_FAST_SQRT(x) __builtin_amdgcn_sqrtf(x)

// ---- sorted register list -------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void list_insert(float (&d)[L], float v) {
#pragma unroll
    for (int t = L - 1; t >= 1; --t) d[t] = __builtin_amdgcn_fmed3f(d[t - 1], v, d[t]);
    d[0] = fminf(d[0], v);
}

// visit every record of the Chebyshev ring block [c-R, c+R]^3 (clamped to the grid)
template <typename F>
__device__ __forceinline__ void scan_block(const GridView &gv, int cx, int cy, int cz, int R, F &&f) {
    const GridGeom &g = gv.g;
    const int x0 = max(cx - R, 0), x1 = min(cx + R, g.gx - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, g.gy - 1);
    const int z0 = max(cz - R, 0), z1 = min(cz + R, g.gz - 1);
    for (int z = z0; z <= z1; ++z) {
        for (int y = y0; y <= y1; ++y) {
            const uint32_t row = ((uint32_t)z * g.gy + y) * g.gx;
            const uint32_t s = gv.cell_start[row + x0], e = gv.cell_start[row + x1 + 1];
            // the record of step i + 1 is requested before step i is evaluated (the padding behind the array makes pts[e] readable):
            // 1-3 % (k = 10 / 16 / 32: 410 -> 398, 535 -> 532, 1048 -> 1029 us)
            // four records requested together (reads past the span stay inside the padded array and are not visited): 143 -> 133 us
            // on a 24 k-point frame (most SIMDs hold one wave there: its dependent round trips are the kernel's time), 520 -> 512 us
            // at 1 M points (one record ahead: 532)
            for (uint32_t j = s; j < e; j += 4) {
                const float4 c0 = gv.pts[j], c1 = gv.pts[j + 1], c2 = gv.pts[j + 2], c3 = gv.pts[j + 3];
                f(j, c0);
                if (j + 1 < e) f(j + 1, c1);
                if (j + 2 < e) f(j + 2, c2);
                if (j + 3 < e) f(j + 3, c3);
            }
        }
    }
}

// distance from q to the box of cell index c along one axis, shaved by the cell-assignment fuzz
// `ext`: the grid's box is clamped (GridGeom::clamped): the first / last cell of the axis (c == 0 / c == last) also
// holds the points beyond the box, so it has no face on that side
template <bool EXT>
__device__ __forceinline__ float axis_gap_n(float q, float mn, float h, int c, int last) {
    const float lo = mn + (float)c * h, hi = lo + h;
    float a = lo - q, b = q - hi;
    if (EXT) {
        a = (c == 0) ? -INFINITY : a;
        b = (c == last) ? -INFINITY : b;
    }
    return fmaxf(fmaxf(a, b) - 2e-3f * h, 0.0f);
}

// visit the records of the cells of block [c-R, c+R]^3 that (a) lie outside block [c-Rin, c+Rin]^3
// (Rin < 0: none excluded) and (b) whose box is within sqrt(lim) of q (ball pruning).  Returns
// whether any cell qualified.
// LIVE: after every row the limit is re-read from *live, which the visitor keeps up to date (a growing block scanned
// with an infinite limit starts pruning as soon as the list is full).
template <bool EXT, bool LIVE = false, typename F>
__device__ __forceinline__ bool scan_pruned(const GridView &gv, const float4 &q, int cx, int cy, int cz, int Rin, int R,
                                            float lim, F &&f, const float *live = nullptr, uint32_t *rowtag = nullptr) {
    const GridGeom &g = gv.g;
    const int x0 = max(cx - R, 0), x1 = min(cx + R, g.gx - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, g.gy - 1);
    const int z0 = max(cz - R, 0), z1 = min(cz + R, g.gz - 1);
    bool touched = false;
    for (int z = z0; z <= z1; ++z) {
        const float gz = axis_gap_n<EXT>(q.z, g.minz, g.h, z, g.gz - 1);
        for (int y = y0; y <= y1; ++y) {
            const float gy = axis_gap_n<EXT>(q.y, g.miny, g.h, y, g.gy - 1);
            const float rg = gy * gy + gz * gz;
            if (rg > lim) continue;
            const bool inner_row = (abs(z - cz) <= Rin) && (abs(y - cy) <= Rin);
            int xa = x0, xb = x1;
            {
                // x window the ball reaches, in closed form: cells whose box is within sqrt(lim - rg) of q.x (slack = the
                // cell-assignment fuzz twice, so never a cell too few; a clamped grid's boundary cells are open on the outer side,
                // which the same bounds cover).  The exact cell-by-cell trim this replaces cost more than the one or two extra
                // cells it saved: 515 -> 461 us at 1 M points, 133 -> 106 us on a 24 k-point frame.
                const float r = TC_FAST_SQRT(fmaxf(lim - rg, 0.0f)) + 4e-3f * g.h;
                const float fa = fminf(fmaxf((q.x - r - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
                const float fb = fmaxf(fminf((q.x + r - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
                xa = max(xa, (int)fa);
                xb = min(xb, (int)fb);
            }
            if (xa > xb) continue;
            const uint32_t row = ((uint32_t)z * g.gy + y) * g.gx;
            if (rowtag) *rowtag = (uint32_t)((((z - cz + 3) & 7) << 3) | ((y - cy + 3) & 7)) << 6;     // knn_tagged: the row's code (|dz|, |dy| <= 3 there)
            auto span = [&](int a, int b) {
                if (a > b) return;
                touched = true;
                const uint32_t s = gv.cell_start[row + a], e = gv.cell_start[row + b + 1];
                for (uint32_t j = s; j < e; j += 4) {
                    const float4 c0 = gv.pts[j], c1 = gv.pts[j + 1], c2 = gv.pts[j + 2], c3 = gv.pts[j + 3];
                    f(j, c0);
                    if (j + 1 < e) f(j + 1, c1);
                    if (j + 2 < e) f(j + 2, c2);
                    if (j + 3 < e) f(j + 3, c3);
                }
            };
            if (!inner_row) span(xa, xb);
            else { span(xa, min(xb, cx - Rin - 1)); span(max(xa, cx + Rin + 1), xb); }   // only the cells outside the inner block
            if (LIVE) lim = *live;
        }
    }
    return touched;
}


// dev build (-DTC_NSTATS): lanes served / fallen back by the tagged-key path, printed per launch
#ifdef TC_NSTATS
__device__ unsigned long long g_nstats[16];     // 0 lanes, 1 served, 2 needs a ring beyond 3 / a span beyond 65535 records, 3 a check failed, 6 waves, 7 waves with a fallback lane
#define TC_NSTAT(i, v) atomicAdd(&g_nstats[i], (unsigned long long)(v))
#else
#define TC_NSTAT(i, v) do { } while (0)
#endif

// ---- tagged-key k-NN (round 3): the neighbour's position rides in the low bits of its list key -----------------------------
// The register-list path scans the neighbourhood twice: once for the sorted distances, once more to find WHICH records they
// were (the collect pass: 41 % of a wave's time at k = 16).  Here the list holds 32-bit keys = squared-distance bits with the low
// 12 mantissa bits replaced by a tag: 6 bits row code ((dz + 3) << 3 | (dy + 3): rings <= 3) and the record's position modulo 64.
// Positive floats order like unsigned integers, so the list is a v_med3_u32 chain; the keys' order is the order of the distances
// truncated to 11 mantissa bits.  After the scan a key names its record: the row from the code, the position = the one in the
// row's 7-cell window that is congruent to the tag (windows of more than 64 records: fallback); the exact distances are computed
// again from the records (19 gathers instead of a second scan), entries whose truncated distances collide are put in exact
// (distance, position) order by counting inversions among neighbours within three places (runs of more than four equal truncated
// values: fallback), and the list is PROVEN to hold the k + 1 nearest: every record that is not in it has a truncated distance
// >= the last key's, so it suffices that the exact (k + 1)-th distance lies below the last key's truncated value -- with two
// spare entries (L = k + 3) that fails for ~1e-5 of uniform points (three consecutive order statistics within 5e-4 relative; measured:
// 5 - 15 lanes per million).
// Exactness rule and ring-3 continuation as in the register-list path, judged against the truncation's upper bound.  A lane that
// fails any check returns false and runs the register-list path: same bits either way.
constexpr uint32_t kTagMask = 0xFFFu, kKeyInf = 0x7f800000u;
typedef float nf32x3 __attribute__((ext_vector_type(3)));
// raw buffer descriptor over the record array: 32-bit byte offsets per lane, the four records of a step share one offset register
// (launch_normals takes this path only below 2^28 records)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t nrm_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xFFFFFFFF, 0x00020000);
}
__device__ __forceinline__ uint32_t tag_key(const nf32x3 &c, const float4 &q, uint32_t rowtag, uint32_t j) {
    // two bit-field inserts (the compiler's own choice for the C expression is and + and + or3)
    uint32_t t, key;
    asm("v_bfi_b32 %0, 63, %1, %2" : "=v"(t) : "v"(j), "v"(rowtag));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(key) : "s"(kTagMask), "v"(t), "v"(__float_as_uint(d2_nc(c.x, c.y, c.z, q.x, q.y, q.z))));
    return key;
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {          // -> v_med3_u32
    const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
    const uint32_t t = mx < c ? mx : c;
    return mn > t ? mn : t;
}
// (Round 6: the same chain on v_med3_f32 -- the keys are bit patterns of non-negative floats, which order like their bits; needs +inf
// instead of 0xFFFFFFFF as the no-op key and f32 denormals kept -- is bit-identical and NOT faster: 273.4 vs 272.2 us.  On gfx950
// v_med3_f32 / v_min_f32 / v_max_f32 issue at the same half rate as v_med3_u32, ~4 cycles per wave64 instruction at any occupancy;
// only f32 add / mul / fma and integer add / and / or / xor run at ~2.5: profiles/r06_valu_issue_price.txt.  -DTC_KEYS_F32 keeps the A/B.)
#ifdef TC_KEYS_F32
constexpr uint32_t kKeyNop = 0x7f800000u;
template <int L>
__device__ __forceinline__ void list_insert_u(uint32_t (&d)[L], uint32_t v) {
    const float fv = __uint_as_float(v);
#pragma unroll
    for (int t = L - 1; t >= 1; --t) d[t] = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(d[t - 1]), fv, __uint_as_float(d[t])));
    d[0] = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(d[0]), fv, 0.0f));      // min(a, v) = med3(a, v, 0) for a, v >= 0
}
#else
constexpr uint32_t kKeyNop = 0xFFFFFFFFu;
template <int L>
__device__ __forceinline__ void list_insert_u(uint32_t (&d)[L], uint32_t v) {
#pragma unroll
    for (int t = L - 1; t >= 1; --t) d[t] = umed3(d[t - 1], v, d[t]);
    d[0] = d[0] < v ? d[0] : v;
}
#endif

// FLAT (round 3, third form): the lockstep row walk -- a row costs the wave its longest span, ~214 candidate slots per lane for a mean
// need of 80 at k = 16 -- becomes three groups of rows (the central 3 x 3, then the outer 16 in two halves, nearest first): the
// row logic of a group runs converged for all lanes (windows judged against the limit the list holds when the group starts, the
// cell_start pairs of all its rows in flight together), the non-empty spans go to a per-lane LDS list, and ONE flattened loop
// walks them -- a lane moves to its next span when its current one ends, so a group costs the wave the longest SUM of spans
// (simulated on the uniform cloud: ~155 slots).  The loop body is branch-free: a slot beyond its span inserts the key
// 0xFFFFFFFF (kKeyNop: a no-op for the list), the next span is prefetched from LDS at the top of every step.
// records per step of the flattened walk: 2 since round 4 (4 before: 304 -> 291 us at 1 M points / k = 16, 227 -> 218 at k = 10;
// 3: 292 / 223).  Build with -DTC_FLAT_W=<n> for an A/B (tools/dev/build_variant.sh).
#ifndef TC_FLAT_W
#define TC_FLAT_W 2
#endif
// Round 6, measured and NOT kept as the default: two groups -- the central 3 x 3, then all sixteen outer rows in ONE flattened walk
// (-DTC_FLAT_GROUPS=2).  The statistics (-DTC_NSTATS, profiles/r06_normals_lockstep.txt) said a lane needs 25.4 + 6.8 + 2.8 steps of two
// records where its wave takes 32.5 + 16.4 + 8.4, and one walk over both outer halves should cost one maximum instead of the sum of two.
// It does not: the merged walk takes 25.0 steps (16.4 + 8.4 = 24.8 before) for a need of 10.1 (9.6) -- the lane that is slowest in the
// first half (a sparse neighbourhood, a large 17th distance) is the slowest in the second half too, so the maximum of the sums IS the sum
// of the maxima, the second half loses the limit the first one tightens, and the span list doubles (9.4 instead of 6.1 KB of LDS per
// wave: 17 instead of 20 waves per CU): 281 - 284 us against 267 for three groups with the same row logic.
#ifndef TC_FLAT_GROUPS
#define TC_FLAT_GROUPS 3
#endif
constexpr int kFlatMaxRows = TC_FLAT_GROUPS == 2 ? 16 : 9;
struct FlatRows { int8_t dz[kFlatMaxRows], dy[kFlatMaxRows]; int n; };
#if TC_FLAT_GROUPS == 2
__device__ constexpr FlatRows kFlatRows[2] = {
    {{0, 0, 0, -1, 1, -1, -1, 1, 1}, {0, -1, 1, 0, 0, -1, 1, -1, 1}, 9},
    {{0, 0, -2, 2, -1, -1, 1, 1, -2, -2, 2, 2, -2, -2, 2, 2}, {-2, 2, 0, 0, -2, 2, -2, 2, -1, 1, -1, 1, -2, 2, -2, 2}, 16},
};
#else
__device__ constexpr FlatRows kFlatRows[3] = {
    {{0, 0, 0, -1, 1, -1, -1, 1, 1}, {0, -1, 1, 0, 0, -1, 1, -1, 1}, 9},
    {{0, 0, -2, 2, -1, -1, 1, 1, 0}, {-2, 2, 0, 0, -2, 2, -2, 2, 0}, 8},
    {{-2, -2, 2, 2, -2, -2, 2, 2, 0}, {-1, 1, -1, 1, -2, 2, -2, 2, 0}, 8},
};
#endif
// words of LDS per lane the flattened walk parks its spans in (two per row of the largest group)
constexpr int kFlatSpanWords = 2 * kFlatMaxRows;

template <int L, int BLOCK, bool EXT, bool FLAT = false>
__device__ __forceinline__ bool knn_tagged(const GridView &gv, const NormalParams &prm, uint32_t p, const float4 &q, int cx, int cy, int cz,
                                           float mf, uint32_t *ldsA, uint8_t *ldsB, uint32_t &cnt, int &self_r, float &d1_out
#ifdef TC_PHASE_STAMPS
                                           , unsigned long long (&ph)[8], unsigned long long &tl
#endif
                                           ) {
    const GridGeom &g = gv.g;
    const uint32_t K1 = prm.k + 1;               // <= L - 2 (launch_normals)
    uint32_t d[L];
#pragma unroll
    for (int t = 0; t < L; ++t) d[t] = kKeyInf;
    auto lim_of = [](uint32_t key) { return __uint_as_float(key < kKeyInf ? key : kKeyInf); };     // a key read as a float bounds its distance from below within 5e-4
    // pruning limit of the block scan: the last key -- or, when k + 1 = L - 2 (k = 16 with L = 19), the (k+1)-th key's upper bound:
    // a record beyond it cannot be among the k + 1 nearest, and the two spare entries only have to bound the records that WERE
    // visited and rejected (the certificate below)
    const bool tight = K1 + 2u == (uint32_t)L;
    auto lim_hi = [](uint32_t key) { return key < kKeyInf ? __uint_as_float(key | kTagMask) : INFINITY; };
    const __amdgpu_buffer_rsrc_t pt_rsrc = nrm_rsrc(gv.pts);
    auto visit = [&](uint32_t j, const float4 &c, uint32_t rowtag) {
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        list_insert_u<L>(d, (__float_as_uint(v) & ~kTagMask) | rowtag | (j & 63u));
    };
    int R = 2;
    if constexpr (FLAT) {
        float live0 = INFINITY;
        bool fail = false;
        // Once per point (round 6; the row logic was 20 % of a wave's time, profiles/r06_normals_phases.txt): the squared gaps of the five
        // z and the five y offsets -- a row's rg is the sum of two of them, the same two products and the same sum as before --, the
        // point's own row, and the unpruned x window.  A row outside the grid keeps the point's own row for its (unused) reads.
        float gz2[5], gy2[5];
        bool zin[5], yin[5];
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int z = cz + d - 2, y = cy + d - 2;
            zin[d] = z >= 0 && z < g.gz;
            yin[d] = y >= 0 && y < g.gy;
            const float gzv = axis_gap_n<EXT>(q.z, g.minz, g.h, zin[d] ? z : cz, g.gz - 1);
            const float gyv = axis_gap_n<EXT>(q.y, g.miny, g.h, yin[d] ? y : cy, g.gy - 1);
            gz2[d] = gzv * gzv;
            gy2[d] = gyv * gyv;
        }
        const uint32_t row0 = ((uint32_t)cz * g.gy + cy) * g.gx;
        const int xlo = max(cx - 2, 0), xhi = min(cx + 2, g.gx - 1);
#pragma unroll
        for (int grp = 0; grp < TC_FLAT_GROUPS; ++grp) {
            constexpr int NR = kFlatMaxRows;
            uint32_t ss[NR], ee[NR];
            bool ok[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                ok[i] = false; ss[i] = 0; ee[i] = 0;
                if (i >= kFlatRows[grp].n) continue;
                const int dz = kFlatRows[grp].dz[i], dy = kFlatRows[grp].dy[i];
                const bool inb = zin[dz + 2] && yin[dy + 2];
                int xa = xlo, xb = xhi;
                if (grp == 0) {
                    // (no limit yet: the whole window of every row inside the grid)
                    ok[i] = inb;
                } else {
                    const float rg = gy2[dy + 2] + gz2[dz + 2];
                    const float r = TC_FAST_SQRT(fmaxf(live0 - rg, 0.0f)) + 4e-3f * g.h;
                    const float fa = fminf(fmaxf((q.x - r - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
                    const float fb = fmaxf(fminf((q.x + r - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
                    xa = max(xa, (int)fa);
                    xb = min(xb, (int)fb);
                    ok[i] = inb && !(rg > live0) && xa <= xb;
                    xb = max(xb, xa);
                }
                const int roff = (dz * g.gy + dy) * g.gx;               // (wave-uniform)
                const uint32_t row = inb ? row0 + (uint32_t)roff : row0;
                ss[i] = gv.cell_start[row + xa];
                ee[i] = gv.cell_start[row + xb + 1];
            }
            uint32_t ns = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if (i >= kFlatRows[grp].n) continue;
                const uint32_t len = ok[i] ? ee[i] - ss[i] : 0u;
                fail |= len > 0xFFFFu;
                const uint32_t rowtag = (uint32_t)(((kFlatRows[grp].dz[i] + 3) << 3) | (kFlatRows[grp].dy[i] + 3)) << 6;
                ldsA[(2u * ns) * BLOCK] = ss[i];
                ldsA[(2u * ns + 1u) * BLOCK] = (len & 0xFFFFu) | (rowtag << 16);
                ns += len ? 1u : 0u;
            }
            { uint32_t sink_ = ns; asm volatile("" :: "v"(sink_)); }
            TC_NSTAMP(2);          // (tagged path) row logic of the group: windows, cell_start pairs, spans parked
#ifdef TC_NSTATS
            {   // lock-step statistics of the flattened walk: steps this lane needs in the group, summed / squared / maximum over the wave
                uint32_t need = 0;
#pragma unroll
                for (int i = 0; i < NR; ++i) { if (i >= kFlatRows[grp].n) continue; const uint32_t len = ok[i] ? ee[i] - ss[i] : 0u; need += (len + TC_FLAT_W - 1u) / TC_FLAT_W; }
                uint32_t mx = need;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
                TC_NSTAT(8 + grp, need);                              // steps needed, per group
                if ((threadIdx.x & 63) == 0) TC_NSTAT(11 + grp, mx);  // steps the wave takes (its slowest lane), per group
            }
#endif
            uint32_t k = 0;
            uint32_t j = 0, e = 0, tag = 0;
            {
                const uint32_t w1 = ldsA[0], w2 = ldsA[BLOCK];
                j = ns ? w1 : 0u;
                e = ns ? w1 + (w2 & 0xFFFFu) : 0u;
                tag = w2 >> 16;
            }
            // (if + do-while: as a plain while loop the compiler copies the whole list at the loop header, 19 v_mov per step)
            if (__any((int)(j < e))) do {
                const uint32_t kn = min(k + 1u, (uint32_t)(NR - 1));
                const uint32_t n1 = ldsA[(2u * kn) * BLOCK], n2 = ldsA[(2u * kn + 1u) * BLOCK];
                const uint32_t o = j << 4;
                // TC_FLAT_W records per step.  A span of len records costs ceil(len / W) * W slots: with ~5 records per span (five cells
                // of ~1 point) W = 4 wastes a third of the slots, W = 2 a tenth -- against ~15 instructions of step overhead (87 per step
                // of two slots: 36 v_med3 for the list, 12 for the distances, 8 for keys and validity)
#pragma unroll
                for (int w = 0; w < TC_FLAT_W; ++w) {
                    const nf32x3 cw = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 16u * (uint32_t)w, 0, 0));
                    const uint32_t kw = tag_key(cw, q, tag, j + (uint32_t)w);
                    list_insert_u<L>(d, j + (uint32_t)w < e ? kw : kKeyNop);
                }
                j += (uint32_t)TC_FLAT_W;
                const bool adv = j >= e;
                k += adv ? 1u : 0u;
                const bool more = k < ns;
                j = adv ? (more ? n1 : 0u) : j;
                tag = adv ? (n2 >> 16) : tag;
                e = adv ? (more ? n1 + (n2 & 0xFFFFu) : 0u) : e;
            } while (__any((int)(j < e)));
            { uint32_t sink_ = d[L - 1]; asm volatile("" :: "v"(sink_)); }
            TC_NSTAMP(3);          // (tagged path) the flattened walk of the group
            // (three times per lane: the (k+1)-th key by a select over the list is affordable here whatever k is)
            {
                uint32_t kk = d[0];
#pragma unroll
                for (int t = 1; t < L; ++t) kk = ((uint32_t)t == prm.k) ? d[t] : kk;
                live0 = lim_hi(kk);
            }
        }
        if (fail) { TC_NSTAT(2, 1); return false; }
    } else {
        // the ring-2 block centre-out, pruned against the list's last key as soon as the list is full (normals_point's list pass)
        float live0 = INFINITY;
        for (int iz = 0; iz < 5; ++iz) {
            const int mz = (iz + 1) >> 1, dz = (iz & 1) ? -mz : mz, z = cz + dz;
            if (z < 0 || z >= g.gz) continue;
            const float gzv = axis_gap_n<EXT>(q.z, g.minz, g.h, z, g.gz - 1);
            for (int iy = 0; iy < 5; ++iy) {
                const int my = (iy + 1) >> 1, dy = (iy & 1) ? -my : my, y = cy + dy;
                if (y < 0 || y >= g.gy) continue;
                const float gyv = axis_gap_n<EXT>(q.y, g.miny, g.h, y, g.gy - 1);
                const float rg = gyv * gyv + gzv * gzv;
                if (rg > live0) continue;
                int xa = max(cx - 2, 0), xb = min(cx + 2, g.gx - 1);
                const float r = TC_FAST_SQRT(fmaxf(live0 - rg, 0.0f)) + 4e-3f * g.h;
                const float fa = fminf(fmaxf((q.x - r - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
                const float fb = fmaxf(fminf((q.x + r - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
                xa = max(xa, (int)fa);
                xb = min(xb, (int)fb);
                if (xa > xb) continue;
                const uint32_t row = ((uint32_t)z * g.gy + y) * g.gx;
                const uint32_t rowtag = (uint32_t)(((dz + 3) << 3) | (dy + 3)) << 6;
                const uint32_t s = gv.cell_start[row + xa], e = gv.cell_start[row + xb + 1];
                for (uint32_t j = s; j < e; j += 4) {
                    const uint32_t o = j << 4;
                    const nf32x3 c0 = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o, 0, 0));
                    const nf32x3 c1 = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 16u, 0, 0));
                    const nf32x3 c2 = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 32u, 0, 0));
                    const nf32x3 c3 = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, o + 48u, 0, 0));
                    list_insert_u<L>(d, tag_key(c0, q, rowtag, j));
                    if (j + 1 < e) list_insert_u<L>(d, tag_key(c1, q, rowtag, j + 1u));
                    if (j + 2 < e) list_insert_u<L>(d, tag_key(c2, q, rowtag, j + 2u));
                    if (j + 3 < e) list_insert_u<L>(d, tag_key(c3, q, rowtag, j + 3u));
                }
                live0 = tight ? lim_hi(d[L - 3]) : lim_of(d[L - 1]);
            }
        }
    }
    for (;;) {
        uint32_t kk = d[0];
#pragma unroll
        for (int t = 1; t < L; ++t) kk = ((uint32_t)t == prm.k) ? d[t] : kk;
        const float tau = kk >= kKeyInf ? INFINITY : __uint_as_float(kk | kTagMask);        // >= the exact (k+1)-th distance
        const bool covers = (cx - R <= 0) && (cx + R >= g.gx - 1) && (cy - R <= 0) && (cy + R >= g.gy - 1) &&
                            (cz - R <= 0) && (cz + R >= g.gz - 1);
        const float bound = ((float)R + mf - 2e-3f) * g.h;
        if (covers || tau <= bound * bound) break;
        if (R >= 3) { TC_NSTAT(2, 1); return false; }         // the row code covers rings <= 3
        if (tau != INFINITY && (int)fminf(ceilf(sqrtf(tau) * g.inv_h - mf + 0.01f), 1.0e9f) > 3) { TC_NSTAT(2, 1); return false; }
        const bool growing = tau == INFINITY;
        R = 3;
        float live_lim = tau;
        uint32_t rowtag = 0;
        const bool touched = scan_pruned<EXT, true>(gv, q, cx, cy, cz, 2, 3, live_lim, [&](uint32_t j, const float4 &c) {
            visit(j, c, rowtag);
            if (growing) live_lim = lim_of(d[L - 1]);
        }, &live_lim, &rowtag);
        if (!touched) break;
    }
    { uint32_t sink_ = d[L - 1]; asm volatile("" :: "v"(sink_)); }
    TC_NSTAMP(1);                  // (tagged path) exactness rule + ring-3 continuation
    // ---- the keys name their records ----
    bool bad = false;
#pragma unroll
    for (int t = 0; t + 4 < L; ++t) bad |= d[t + 4] < kKeyInf && (d[t] & ~kTagMask) == (d[t + 4] & ~kTagMask);      // five equal truncated distances
    const uint32_t last_floor = d[L - 1] & ~kTagMask;
    const bool full = d[L - 1] < kKeyInf;
    uint32_t nv = 0;
#pragma unroll
    for (int t = 0; t < L; ++t) { ldsA[t * BLOCK] = d[t]; nv += d[t] < kKeyInf ? 1u : 0u; }
    cnt = min(K1, nv);
    const int x0 = max(cx - 3, 0), x1 = min(cx + 3, g.gx - 1);
    // The list leaves the registers (a decode step unrolled over all L entries keeps every gather in flight: 191 VGPRs): the keys
    // are parked in ldsA, a rolled loop takes them four at a time -- key -> row window -> position -> record -> exact distance --
    // and the exact (distance, position) order comes out of a sliding window: a key's place is off by at most three (runs of <= 4
    // equal truncated distances), so an entry's rank is final once three later entries have been compared with it; its position
    // then goes to ldsA[rank] (never ahead of the keys still to be read).
    float wv0 = -INFINITY, wv1 = -INFINITY, wv2 = -INFINITY;         // window: 0 = oldest
    uint32_t wj0 = 0, wj1 = 0, wj2 = 0, wr0 = 0, wr1 = 0, wr2 = 0, wt0 = 0xFFu, wt1 = 0xFFu, wt2 = 0xFFu;
    float vk = INFINITY, v1 = INFINITY;
    self_r = -1;
    auto push = [&](float vn, uint32_t jn, uint32_t tn) {
        uint32_t rn = tn;
        const uint32_t c2 = (vn < wv2 || (vn == wv2 && jn < wj2)) ? 1u : 0u;
        const uint32_t c1 = (vn < wv1 || (vn == wv1 && jn < wj1)) ? 1u : 0u;
        const uint32_t c0 = (vn < wv0 || (vn == wv0 && jn < wj0)) ? 1u : 0u;
        rn -= c0 + c1 + c2;
        wr0 += c0; wr1 += c1; wr2 += c2;
        if (wt0 != 0xFFu) {                      // the oldest entry's rank is final
            vk = (wr0 + 1u == K1) ? wv0 : vk;
            v1 = (wr0 == 1u) ? wv0 : v1;
            if (wj0 == p && wr0 < cnt) self_r = (int)wr0;
#ifdef TC_TAG_NOBATCH
            ldsB[wr0 * BLOCK] = (uint8_t)wr0;
#endif
            ldsA[wr0 * BLOCK] = wj0;             // rank order (ranks <= the entry last read from ldsA: its key is in a register by now)
        }
        wv0 = wv1; wj0 = wj1; wr0 = wr1; wt0 = wt1;
        wv1 = wv2; wj1 = wj2; wr1 = wr2; wt1 = wt2;
        wv2 = vn; wj2 = jn; wr2 = rn; wt2 = tn;
    };
    // (requesting the keys and row windows of the NEXT four entries before the current four are resolved was measured: no gain at
    // k = 10, and at k = 16 the extra live registers spill -- 382 instead of 325 us)
    uint32_t keyN[4], sN[4], eN[4];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            keyN[i] = (t0 + i < L) ? ldsA[(t0 + i) * BLOCK] : kKeyInf;
            const bool valid = keyN[i] < kKeyInf;
            const uint32_t tag = keyN[i] & kTagMask;
            const int z = valid ? cz + (int)(tag >> 9) - 3 : cz, y = valid ? cy + (int)((tag >> 6) & 7u) - 3 : cy;
            const uint32_t row = ((uint32_t)z * g.gy + y) * g.gx;
            sN[i] = gv.cell_start[row + x0];
            eN[i] = gv.cell_start[row + x1 + 1];
        }
    };
#pragma unroll 1
    for (int t0 = 0; t0 < L; t0 += 4) {
        fetch(t0);
        uint32_t key[4], jx[4];
        bool valid[4];
        nf32x3 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            key[i] = keyN[i];
            valid[i] = key[i] < kKeyInf;
            bad |= valid[i] && (eN[i] - sN[i] > 64u);
            const uint32_t j = sN[i] + (((key[i] & 63u) - sN[i]) & 63u);
            jx[i] = (valid[i] && j < eN[i]) ? j : p;
            c[i] = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, jx[i] << 4, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float ve = d2_nc(c[i].x, c[i].y, c[i].z, q.x, q.y, q.z);
            bad |= valid[i] && ((__float_as_uint(ve) & ~kTagMask) != (key[i] & ~kTagMask));
            if (t0 + i < L) {
                push(valid[i] ? ve : INFINITY, valid[i] ? jx[i] : 0xFFFFFFFFu, (uint32_t)(t0 + i));
            }
        }
    }
    push(INFINITY, 0xFFFFFFFFu, 0xFFu); push(INFINITY, 0xFFFFFFFFu, 0xFFu); push(INFINITY, 0xFFFFFFFFu, 0xFFu);
    // every record outside the list has a truncated distance >= the last key's
    if (full) bad |= !(vk < __uint_as_float(last_floor));
    { float sink_ = vk + v1; asm volatile("" :: "v"(sink_)); }
    TC_NSTAMP(4);                  // (tagged path) decode: keys -> records -> exact order + certificate
    d1_out = v1;
    if (bad) TC_NSTAT(3, 1); else TC_NSTAT(1, 1);
    TC_NSTAT(0, 1);
    return !bad;
}

template <int L, int BLOCK, bool RADIUS, bool EXT, int CAP = 0>
__device__ __forceinline__ void normals_point(const GridView &gv, const NormalParams &prm, uint32_t p,
                                              float *__restrict__ out6, uint32_t *ldsA, uint8_t *ldsB
#ifdef TC_PHASE_STAMPS
                                              , unsigned long long (&ph)[8], unsigned long long &tl
#endif
                                              ) {
    const GridGeom &g = gv.g;
    const float4 q = gv.pts[p];
    const uint32_t orig = __float_as_uint(q.w);
    const uint32_t K1 = prm.k + 1;
    if (p >= gv.cell_start[g.ncell]) {
        // A point with a NaN / infinite coordinate (the bucket behind the last cell; its record holds placeholder coordinates).
        // The reference's PCA of such a neighbourhood is NaN throughout: `norm > 1e-6` is false -> the default normal (0, 0, 1)
        // (normals.rs:197-202), and the orientation test `dot < 0` is false for NaN (:216).  Finite points never see these
        // points as neighbours here (the reference's kd-tree places them wherever its NaN comparisons fall).
        if (prm.sorted_nrm) prm.sorted_nrm[p] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
        if (prm.vor_out) prm.vor_out[p] = make_float4(q.x, q.y, q.z, 0.0f);
        if (out6) {
            float *o = out6 + 6 * (size_t)(prm.slice_out ? p - prm.p_begin : orig);
            o[0] = prm.xyz[3 * (size_t)orig]; o[1] = prm.xyz[3 * (size_t)orig + 1]; o[2] = prm.xyz[3 * (size_t)orig + 2];
            o[3] = 0.0f; o[4] = 0.0f; o[5] = 1.0f;
        }
        return;
    }
    const int cx = cell_coord(q.x, g.minx, g.inv_h, g.gx);
    const int cy = cell_coord(q.y, g.miny, g.inv_h, g.gy);
    const int cz = cell_coord(q.z, g.minz, g.inv_h, g.gz);
    // distance from the query to the nearest face of its own cell
    float fx = (q.x - g.minx) * g.inv_h - (float)cx, fy = (q.y - g.miny) * g.inv_h - (float)cy,
          fz = (q.z - g.minz) * g.inv_h - (float)cz;
    float mf = fminf(fminf(fminf(fx, 1.0f - fx), fminf(fy, 1.0f - fy)), fminf(fz, 1.0f - fz));
    mf = fmaxf(mf, 0.0f);

    // (inside a tagged-key kernel this is the fallback path: its list needs k + 1 entries, not the tagged list's k + 3)
    constexpr int LO = (CAP < 0) ? L - 2 : L;
    float d[LO];
    float tau = INFINITY;
    int R = prm.R0;
    // radius mode (normals.rs:141-146): neighbours = every record within `radius` except the query
    // itself; their moments about the query are accumulated in f64 during the same scan
    // (order independent; the reference's f32 sums in ascending-distance order differ by rounding only)
    const float r2 = prm.radius * prm.radius;
    uint32_t cnt_r = 0;
    double s1x = 0, s1y = 0, s1z = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    bool use_radius = false;
#pragma unroll
    for (int t = 0; t < LO; ++t) d[t] = INFINITY;
    bool have = false;
    uint32_t cnt = 0;
    int self_r = -1;
    if constexpr (CAP < 0 && !RADIUS) {
        // the tagged-key path (knn_tagged): on success ldsA / ldsB hold the k+1 nearest and their ranks, d[1] the distance to the
        // nearest other record; a lane it cannot serve runs the register-list path below
        TC_NSTAMP(0);
        // Where it pays is decided per index, on the device (no host round trip): the tagged path wins where the cells are
        // filled like a volume (uniform 1 M points, k = 16: 435 -> 315 us) and loses on surfaces and dense cells -- few rows of a
        // block hold points there, the old collect pass is cheap, and row windows of more than 64 records send lanes to the
        // fallback (TUM-shaped 1 M: 407 -> 440 us, a 120 k-point LiDAR sweep on an unadapted grid 190 -> 355).  The index
        // build leaves the number of occupied cells in front of the prefix sums: at least 20 % of the cells occupied (a uniform
        // cloud: 40 - 76 % depending on k; an adapted surface grid: 3 - 6 %) and at most 4 points per occupied cell = volumetric.
        bool try_tag = true;
        if (prm.tag_policy) {
            const uint32_t occ = gv.cell_start[-(int)kCellStartFront];
            try_tag = (unsigned long long)occ * 5ull >= (unsigned long long)g.ncell && (unsigned long long)g.n <= (unsigned long long)occ * 4ull;
        }
        if (try_tag) {
            float d1 = INFINITY;
            have = knn_tagged<L, BLOCK, EXT, (CAP < -1)>(gv, prm, p, q, cx, cy, cz, mf, ldsA, ldsB, cnt, self_r, d1
#ifdef TC_PHASE_STAMPS
                                                         , ph, tl
#endif
                                                         );
            if (have) d[1] = d1;
            TC_NSTAMP(1);
        }
#ifdef TC_NSTATS
        if ((threadIdx.x & 63) == 0) TC_NSTAT(6, 1);
        if (__any((int)!have) && (threadIdx.x & 63) == 0) TC_NSTAT(7, 1);
#endif
    }
    auto visit1 = [&](uint32_t j, const float4 &c) {
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        list_insert<LO>(d, v);
        if (RADIUS && j != p && v <= r2) {
            const double dx = (double)c.x - (double)q.x, dy = (double)c.y - (double)q.y, dz = (double)c.z - (double)q.z;
            ++cnt_r;
            s1x += dx; s1y += dy; s1z += dz;
            sxx = fma(dx, dx, sxx); sxy = fma(dx, dy, sxy); sxz = fma(dx, dz, sxz);
            syy = fma(dy, dy, syy); syz = fma(dy, dz, syz); szz = fma(dz, dz, szz);
        }
    };
#define TC_KTH(OUT)                                                                 \
    do {                                                                            \
        float tk_ = d[0];                                                           \
        _Pragma("unroll") for (int t = 1; t < LO; ++t) tk_ = ((uint32_t)t == prm.k) ? d[t] : tk_; \
        (OUT) = tk_;                                                                \
    } while (0)
    // ring R0: the whole block (no bound known yet)
    TC_NSTAMP(0);
    if (!have) {
    {
        // the block's rows centre-out (dz = 0, -1, +1, -2, +2; inside, dy likewise): the list is full after the central rows and every
        // later row is judged (skipped, or trimmed in x in closed form) against a real limit -- 460 -> 433 us at 1 M points against
        // the plain block scan (the same scan in z-major order: 460; ring 1 unpruned + pruned ring-2 shell: 479; the next row's
        // cell_start pair requested one row ahead, i.e. the row logic inside a per-lane loop: 516)
        const float need0 = RADIUS ? fmaxf(r2, 0.0f) : 0.0f;
        float live0 = INFINITY;
        const int W = 2 * R + 1;
        for (int iz = 0; iz < W; ++iz) {
            const int mz = (iz + 1) >> 1, z = cz + ((iz & 1) ? -mz : mz);
            if (z < 0 || z >= g.gz) continue;
            const float gzv = axis_gap_n<EXT>(q.z, g.minz, g.h, z, g.gz - 1);
            for (int iy = 0; iy < W; ++iy) {
                const int my = (iy + 1) >> 1, y = cy + ((iy & 1) ? -my : my);
                if (y < 0 || y >= g.gy) continue;
                const float gyv = axis_gap_n<EXT>(q.y, g.miny, g.h, y, g.gy - 1);
                const float rg = gyv * gyv + gzv * gzv;
                if (rg > live0) continue;
                int xa = max(cx - R, 0), xb = min(cx + R, g.gx - 1);
                const float r = TC_FAST_SQRT(fmaxf(live0 - rg, 0.0f)) + 4e-3f * g.h;
                const float fa = fminf(fmaxf((q.x - r - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
                const float fb = fmaxf(fminf((q.x + r - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
                xa = max(xa, (int)fa);
                xb = min(xb, (int)fb);
                if (xa > xb) continue;
                const uint32_t row = ((uint32_t)z * g.gy + y) * g.gx;
                const uint32_t s = gv.cell_start[row + xa], e = gv.cell_start[row + xb + 1];
                for (uint32_t j = s; j < e; j += 4) {
                    const float4 c0 = gv.pts[j], c1 = gv.pts[j + 1], c2 = gv.pts[j + 2], c3 = gv.pts[j + 3];
                    visit1(j, c0);
                    if (j + 1 < e) visit1(j + 1, c1);
                    if (j + 2 < e) visit1(j + 2, c2);
                    if (j + 3 < e) visit1(j + 3, c3);
                }
                live0 = fmaxf(d[LO - 1], need0);
            }
        }
    }
    { float sink_ = d[LO - 1]; asm volatile("" :: "v"(sink_)); }
    TC_NSTAMP(1);
    for (;;) {
        TC_KTH(tau);
        const bool covers = (cx - R <= 0) && (cx + R >= g.gx - 1) && (cy - R <= 0) && (cy + R >= g.gy - 1) &&
                            (cz - R <= 0) && (cz + R >= g.gz - 1);
        const float bound = ((float)R + mf - 2e-3f) * g.h;
        // enough radius neighbours (normals.rs:315): the radius set is used, and it is complete iff
        // the ring covers the radius ball; otherwise the k-NN fallback needs the usual tau rule
        use_radius = RADIUS && cnt_r >= prm.k;
        const bool exact = use_radius ? (prm.radius <= bound) : (tau <= bound * bound);
        if (covers || exact) break;
        // not provably exact (grid-boundary points, Poisson tail): continue IN PLACE with the next
        // shell, visiting only the cells the ball of the current bound can reach.  The ball clipped to
        // the grid box is convex and contains the query, so a shell that misses it ends the search.
        // While the list is not full nothing can be pruned anyway, so the block may grow by half its radius at
        // a time instead of one ring: an isolated point D cells from its neighbours pays ~D^2 row visits, not D^3.
        // With a full list the block radius that proves it is known: (R' + mf) h >= sqrt(tau).  Going there in one
        // call visits every row once; ring by ring every call loops over all rows of its block again.
        const int Rin = R;
        const float need2 = RADIUS ? fmaxf(r2, 0.0f) : 0.0f;     // radius mode must also see the whole radius ball
        if (tau == INFINITY) R += max(1, R / 2);
        else R = max(R + 1, (int)fminf(ceilf(sqrtf(fmaxf(tau, need2)) * g.inv_h - mf + 0.01f), 1.0e9f));
        // An isolated point (a far outlier: its neighbours are a hundred extents away; a point in an empty region) would walk
        // thousands of rows -- the whole grid -- through ONE lane: handed to normals_coop_kernel instead, a wave per point.
        if (!RADIUS && prm.hard_list != nullptr && R > kHardRing) {
            prm.hard_list[4u + atomicAdd(&prm.hard_list[0], 1u)] = p;
            return;
        }
        // one call site for both cases (lanes of a wave differ): only the growing lanes refresh their limit
        // one call site (a second, plain one costs 20 registers and sends the list to scratch)
        const bool growing = tau == INFINITY || R > Rin + 1;
        float live_lim = fmaxf(tau, need2);
        const bool touched = scan_pruned<EXT, true>(gv, q, cx, cy, cz, Rin, R, live_lim, [&](uint32_t j, const float4 &c) {
            visit1(j, c);
            // only the growing lanes tighten their limit; the list's last entry bounds the k-th (no dynamic index: that
            // sends the list to scratch here)
            if (growing) live_lim = fmaxf(d[LO - 1], need2);
        }, &live_lim);
        if (!touched) { TC_KTH(tau); use_radius = RADIUS && cnt_r >= prm.k; break; }
    }
    }   // !have

#undef TC_KTH
    TC_NSTAMP(2);
    // d[0] is the query itself (0), d[1] the squared distance to its nearest OTHER record (0 for an exact duplicate: never kept)
    if (prm.vor_out) prm.vor_out[p] = make_float4(q.x, q.y, q.z, 0.25f * 0.9999f * d[1]);
    float nrm_x = 0.0f, nrm_y = 0.0f, nrm_z = 1.0f;
    // Radius mode with a radius set that fits the list (the query + at most L - 1 others): the set is exactly the list's first
    // cnt_r + 1 entries, so it goes through the k-NN epilogue with k + 1 := cnt_r + 1 -- neighbours in ascending distance, the
    // query last, f32 centroid / covariance in the reference's order (normals.rs:141-146, :164-177), nalgebra's eigen solve:
    // bit-comparable like the k-NN path.  A larger set keeps the order-free f64 moments + closed form below.
    const bool radius_fit = RADIUS && use_radius && cnt_r + 1u <= (uint32_t)LO;
    uint32_t K1e = K1;
    if (radius_fit) {
        K1e = cnt_r + 1u;
        float tr = d[0];
#pragma unroll
        for (int t = 1; t < LO; ++t) tr = ((uint32_t)t == cnt_r) ? d[t] : tr;
        tau = tr;
    }
    if (RADIUS && use_radius && !radius_fit) {
        const double nn = (double)cnt_r + 1.0;                     // + the query itself (normals.rs:338-340)
        const double mx = s1x / nn, my = s1y / nn, mz = s1z / nn;
        double ex, ey, ez;
        smallest_eigvec_sym3(sxx / nn - mx * mx, sxy / nn - mx * my, sxz / nn - mx * mz, syy / nn - my * my,
                             syz / nn - my * mz, szz / nn - mz * mz, ex, ey, ez);
        const float vx = (float)ex, vy = (float)ey, vz = (float)ez;
        const float mag = sqrtf(vx * vx + vy * vy + vz * vz);
        if (mag > 1e-6f) { nrm_x = vx / mag; nrm_y = vy / mag; nrm_z = vz / mag; }
    } else {
    if (!have) {
    // phase 2: collect the positions of the K1 nearest records
    uint32_t n_lt = 0;
#pragma unroll
    for (int t = 0; t < LO; ++t) n_lt += (d[t] < tau) ? 1u : 0u;
    const uint32_t quota = K1e - min(n_lt, K1e);
    uint32_t ties = 0;
    cnt = 0;
    // rescan only the cells within sqrt(tau) of the query (same visiting order as phase 1).  (Measured instead: the plain,
    // unpruned block scan for lanes that never grew their block: 554 vs 536 us -- the lanes of a wave walk the rows in lockstep
    // either way; a centre-out flattened walk with live pruning, every lane opening its next row inside the candidate loop:
    // 1.2 ms -- the row logic then runs divergently at every step; the same with the rows' spans computed by all lanes together
    // in three batches (9 central rows, then 2 x 8 outer rows judged against the list's last entry), parked in LDS and walked
    // flattened, lanes whose ring-2 block is not enough left to a second launch of this kernel: 648 us + 264 us for the second
    // launch (0.2 % of the points, but a lone wave's 18 k dependent instructions ARE its duration): VALU instructions -17 %, SALU
    // +117 % (exec-mask bookkeeping of the divergent loops): the per-candidate insertion is the cost, not the number of steps;
    // the list pass as ring 1 unpruned + ring-2 shell judged against the list's last entry: 590 vs 497 us.)
    scan_pruned<EXT>(gv, q, cx, cy, cz, -1, R, tau, [&](uint32_t j, const float4 &c) {
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        bool take = v < tau;
        if (!take && v == tau && ties < quota) { take = true; ++ties; }
        if (take && cnt < K1e) { ldsA[cnt * BLOCK] = j; ++cnt; }
    });
    TC_NSTAMP(3);
    // rank -> ascending-distance order (ties keep scan order)
    unsigned long long taken_lo = 0ull, taken_hi = 0ull, taken_x = 0ull;   // bitset over ranks 0..191 (L <= 129)
    auto is_taken = [&](uint32_t r) { return r < 64 ? ((taken_lo >> r) & 1ull) : r < 128 ? ((taken_hi >> (r - 64)) & 1ull) : ((taken_x >> (r - 128)) & 1ull); };
    self_r = -1;
    for (uint32_t e = 0; e < cnt; ++e) {
        const uint32_t j = ldsA[e * BLOCK];
        const float4 c = gv.pts[j];
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        uint32_t r = 0;
#pragma unroll
        for (int t = 0; t < LO; ++t) r += (d[t] < v) ? 1u : 0u;
        while (is_taken(r)) ++r;
        if (r < 64) taken_lo |= 1ull << r; else if (r < 128) taken_hi |= 1ull << (r - 64); else taken_x |= 1ull << (r - 128);
        ldsB[r * BLOCK] = (uint8_t)e;           // rank -> entry index (one byte; the positions stay in ldsA)
        if (j == p) self_r = (int)r;
    }
    }   // !have
    TC_NSTAMP(4);
    // normals.rs:147-153: drop self from the k+1 list (or the last entry when self is not in it)
    const int drop_r = (self_r >= 0) ? self_r : (int)cnt - 1;
    const uint32_t npts = cnt;   // (cnt - 1) neighbours + self
    if (npts >= 3) {
        // centroid (normals.rs:165-169): sequential f32 adds, neighbours ascending then self
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        float cxx = 0.0f, cxy = 0.0f, cxz = 0.0f, cyy = 0.0f, cyz = 0.0f, czz = 0.0f;
        const float nf = (float)npts;
        float mx = 0.0f, my = 0.0f, mz = 0.0f;
        bool batched = false;
#ifndef TC_TAG_NOBATCH
        if constexpr (CAP < 0 && !RADIUS) batched = have;
#endif
        if (batched) {
            // tagged path: ldsA holds the positions in rank order.  The k + 1 records are requested TOGETHER and kept in registers
            // (the list's registers are free by now) for both passes, instead of two loops of LDS -> LDS -> gather round trips
            // (34 dependent gathers were 12 % of a wave's time); the additions keep the reference's order, slots beyond the
            // neighbourhood and the dropped entry are skipped by predicate.
            if constexpr (CAP < 0) {
                constexpr int KM = L - 2;
                const __amdgpu_buffer_rsrc_t pt_rsrc = nrm_rsrc(gv.pts);
                float nx[KM], ny[KM], nz[KM];
#pragma unroll
                for (int t = 0; t < KM; ++t) {
                    const uint32_t j = ((uint32_t)t < cnt) ? ldsA[t * BLOCK] : p;
                    const nf32x3 c = __builtin_bit_cast(nf32x3, __builtin_amdgcn_raw_buffer_load_b96(pt_rsrc, j << 4, 0, 0));
                    nx[t] = c.x; ny[t] = c.y; nz[t] = c.z;
                }
#pragma unroll
                for (int t = 0; t < KM; ++t)
                    if ((uint32_t)t < cnt && t != drop_r) { sx += nx[t]; sy += ny[t]; sz += nz[t]; }
                sx += q.x; sy += q.y; sz += q.z;
                mx = sx / nf; my = sy / nf; mz = sz / nf;
#pragma unroll
                for (int t = 0; t < KM; ++t)
                    if ((uint32_t)t < cnt && t != drop_r) {
                        const float dx = nx[t] - mx, dy = ny[t] - my, dz = nz[t] - mz;
                        cxx += dx * dx; cxy += dx * dy; cxz += dx * dz; cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
                    }
            }
        } else {
        for (uint32_t r = 0; r < cnt; ++r) {
            if ((int)r == drop_r) continue;
            const float4 c = gv.pts[ldsA[(uint32_t)ldsB[r * BLOCK] * BLOCK]];
            sx += c.x; sy += c.y; sz += c.z;
        }
        sx += q.x; sy += q.y; sz += q.z;
        mx = sx / nf; my = sy / nf; mz = sz / nf;
        // covariance (normals.rs:172-177)
        for (uint32_t r = 0; r < cnt; ++r) {
            if ((int)r == drop_r) continue;
            const float4 c = gv.pts[ldsA[(uint32_t)ldsB[r * BLOCK] * BLOCK]];
            const float dx = c.x - mx, dy = c.y - my, dz = c.z - mz;
            cxx += dx * dx; cxy += dx * dy; cxz += dx * dz; cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
        }
        }
        {
            const float dx = q.x - mx, dy = q.y - my, dz = q.z - mz;
            cxx += dx * dx; cxy += dx * dy; cxz += dx * dz; cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
        }
        cxx /= nf; cxy /= nf; cxz /= nf; cyy /= nf; cyz /= nf; czz /= nf;
        { float sink_ = cxx + cxy + cxz + cyy + cyz + czz; asm volatile("" :: "v"(sink_)); }
        TC_NSTAMP(5);
        float e0, e1, e2, x0, y0, z0, x1, y1, z1, x2, y2, z2;
        sym_eigen3_f32(cxx, cxy, cxz, cyy, cyz, czz, e0, e1, e2, x0, y0, z0, x1, y1, z1, x2, y2, z2);          // normals.rs:181
        // first index with the strictly smallest eigenvalue (normals.rs:186-191), its column of q
        float vx = x0, vy = y0, vz = z0, emin = e0;
        if (e1 < emin) { emin = e1; vx = x1; vy = y1; vz = z1; }
        if (e2 < emin) { vx = x2; vy = y2; vz = z2; }
        const float mag = sqrtf(vx * vx + vy * vy + vz * vz);          // normals.rs:197-202
        if (mag > 1e-6f) { nrm_x = vx / mag; nrm_y = vy / mag; nrm_z = vz / mag; }
    }
    }   // k-NN path
    if (prm.orient) {   // normals.rs:208-222
        const float tx = prm.vx - q.x, ty = prm.vy - q.y, tz = prm.vz - q.z;
        const float tn = sqrtf(tx * tx + ty * ty + tz * tz);
        const float ux = tx / tn, uy = ty / tn, uz = tz / tn;
        const float dp = nrm_x * ux + nrm_y * uy + nrm_z * uz;
        if (dp < 0.0f) { nrm_x = -nrm_x; nrm_y = -nrm_y; nrm_z = -nrm_z; }
    }
    { float sink_ = nrm_x + nrm_y + nrm_z; asm volatile("" :: "v"(sink_)); }
    TC_NSTAMP(6);
    if (prm.sorted_nrm) prm.sorted_nrm[p] = make_float4(nrm_x, nrm_y, nrm_z, 0.0f);      // coalesced: lane = position
    if (out6) {
        float *o = out6 + 6 * (size_t)(prm.slice_out ? p - prm.p_begin : orig);
        o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = nrm_x; o[4] = nrm_y; o[5] = nrm_z;
    }
    TC_NSTAMP(7);
}

// ---- cooperative k-NN + PCA: one wave per point (round 3) -------------------------------------------------------------------
// For the points the lane-per-point kernel cannot serve: k_neighbors > 128 (its sorted list lives in registers: 129 entries at
// most; the reference has no cap, normals.rs:17-26) and HARD points, whose neighbours are so far away that one lane would walk the
// whole grid (normals_point hands them over through prm.hard_list).  The wave enumerates the ball of a radius r clipped to the
// grid row by row (rows dealt to the lanes, closed-form x windows), appends (distance bits << 32 | position) of every record within
// r to an LDS buffer, and adapts r until the buffer holds at least k + 1 and at most CAPB records -- then the buffer IS the ball,
// a bitonic sort orders it by (distance, position): the first k + 1 keys are the reference's neighbour list, ties to the lowest
// position like the register-list path.  Lane 0 runs the reference's f32 centroid / covariance / eigen sequence over the
// neighbours' coordinates (parked in LDS by all lanes).  Same bits as normals_point on every point both can serve.
constexpr int kCoopThreads = 256;

// shared scratch of one block of the wave-per-point kernels
template <int CAPB>
struct CoopShared {
    unsigned long long buf[CAPB];
    uint32_t hist[256];         // squared distances of the current ball over (hlo, lim], 256 bins: where an overflowing ball is cut
    uint32_t cnt;
    int bin;
};

// The K1 nearest records of q (any point, inside or outside the grid), as keys (distance bits << 32 | position) sorted ascending in
// sh.buf[0 .. return value): adaptive ball + LDS buffer + bitonic sort (see normals_coop_kernel).  Called by all threads of the block.
template <int CAPB>
__device__ __forceinline__ uint32_t coop_nearest(const GridView &gv, float qx, float qy, float qz, uint32_t K1, uint32_t nfin, CoopShared<CAPB> &sh) {
    const GridGeom &g = gv.g;
    const int tid = threadIdx.x;
    // the radius that certainly holds the whole cloud: the distance to the farthest corner of its (grid) box -- a clamped box
    // has records beyond it: there only the counts end the growth
    const float fxm = fmaxf(fabsf(qx - g.minx), fabsf(qx - g.maxx)), fym = fmaxf(fabsf(qy - g.miny), fabsf(qy - g.maxy)),
                fzm = fmaxf(fabsf(qz - g.minz), fabsf(qz - g.maxz));
    const float r_all = g.clamped ? 3.0e38f : sqrtf(fxm * fxm + fym * fym + fzm * fzm) * 1.001f;
    // the first radius at which the box proper comes into reach of a point outside it; no record INSIDE an exact box is
    // closer than the box (a clamped box has records beyond it, possibly nearer: they fall into the first histogram bin)
    const float bxo = fmaxf(fmaxf(g.minx - qx, qx - g.maxx), 0.0f), byo = fmaxf(fmaxf(g.miny - qy, qy - g.maxy), 0.0f),
                bzo = fmaxf(fmaxf(g.minz - qz, qz - g.maxz), 0.0f);
    const float d_box2 = (bxo * bxo + byo * byo + bzo * bzo) * 0.9999f;
    const float r_box = sqrtf(d_box2) + 2.0f * g.h;
    float r = 2.0f * g.h * cbrtf((float)K1 / 17.0f);
    // The ball is cut by KEY = (distance bits << 32 | position), compared exactly: khi = the largest key admitted to the buffer, klo =
    // a key known to have fewer than K1 records at or below it.  (The cut used to be a squared radius with relative safety factors
    // of 1e-5: a plateau of 1 500 exact duplicates 9e-6 beyond the (k+1)-th neighbour could not be cut off, the buffer overflowed and
    // the neighbours were whichever 512 records arrived first -- fuzz seed 611 case 3568.)
    unsigned long long klo = (!g.clamped && d_box2 > 0.0f) ? ((unsigned long long)__float_as_uint(d_box2) << 32) : 0ull;
    unsigned long long khi = ((unsigned long long)__float_as_uint(r * r) << 32) | 0xFFFFFFFFull;
    uint32_t total = 0;
    for (int guard = 0; guard < 200; ++guard) {
        if (tid == 0) sh.cnt = 0;
        sh.hist[tid] = 0;
        __syncthreads();
        const float lim = __uint_as_float((uint32_t)(khi >> 32));          // every admitted record lies within this squared radius
        r = sqrtf(lim) * 1.000001f;
        // 256 bins over the keys in (klo, khi]: bin = (key - klo - 1) >> sh
        const unsigned long long range = khi - klo;
        const int sh_bits = max(0, 64 - (int)__clzll((long long)(range - 1ull) | 1ll) - 8);
        const float ry = r * 1.0001f + 4e-3f * g.h;
        const int y0 = cell_coord(fminf(fmaxf(qy - ry, g.miny), g.maxy), g.miny, g.inv_h, g.gy), y1 = cell_coord(fminf(fmaxf(qy + ry, g.miny), g.maxy), g.miny, g.inv_h, g.gy);
        const int z0 = cell_coord(fminf(fmaxf(qz - ry, g.minz), g.maxz), g.minz, g.inv_h, g.gz), z1 = cell_coord(fminf(fmaxf(qz + ry, g.minz), g.maxz), g.minz, g.inv_h, g.gz);
        const int ny = y1 - y0 + 1;
        const uint32_t nrows = (uint32_t)ny * (uint32_t)(z1 - z0 + 1);
        auto take = [&](uint32_t j, const float4 &c) {
            const float v = d2_nc(c.x, c.y, c.z, qx, qy, qz);
            const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | j;
            if (key <= khi) {
                const uint32_t slot = atomicAdd(&sh.cnt, 1u);
                if (slot < (uint32_t)CAPB) sh.buf[slot] = key;
                if (key > klo) atomicAdd(&sh.hist[(uint32_t)min((key - klo - 1ull) >> sh_bits, 255ull)], 1u);
            }
        };
        for (uint32_t ri = (uint32_t)tid; ri < nrows; ri += kCoopThreads) {
            const int zz = z0 + (int)(ri / (uint32_t)ny), yy = y0 + (int)(ri % (uint32_t)ny);
            const float gy = g.clamped ? axis_gap_n<true>(qy, g.miny, g.h, yy, g.gy - 1) : axis_gap_n<false>(qy, g.miny, g.h, yy, g.gy - 1);
            const float gz = g.clamped ? axis_gap_n<true>(qz, g.minz, g.h, zz, g.gz - 1) : axis_gap_n<false>(qz, g.minz, g.h, zz, g.gz - 1);
            const float rg = gy * gy + gz * gz;
            if (rg > lim) continue;
            const float rx = sqrtf(fmaxf(lim - rg, 0.0f)) * 1.0001f + 4e-3f * g.h;
            const int xa = (int)fminf(fmaxf((qx - rx - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
            const int xb = (int)fmaxf(fminf((qx + rx - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
            if (xa > xb) continue;
            const uint32_t row = ((uint32_t)zz * g.gy + yy) * g.gx;
            const uint32_t s = gv.cell_start[row + xa], e = gv.cell_start[row + xb + 1];
            for (uint32_t j = s; j < e; j += 4) {          // (reads past the span stay inside the padded array)
                const float4 c0 = gv.pts[j], c1 = gv.pts[j + 1], c2 = gv.pts[j + 2], c3 = gv.pts[j + 3];
                take(j, c0);
                if (j + 1 < e) take(j + 1, c1);
                if (j + 2 < e) take(j + 2, c2);
                if (j + 3 < e) take(j + 3, c3);
            }
        }
        __syncthreads();
        total = sh.cnt;
        if (total > (uint32_t)CAPB) {
            // too many for the buffer: cut at the bin in which the count reaches K1 -- the new range holds the K1-th key and 1/256
            // of the old one; a range of <= 256 keys has one key per bin, the cut is then the K1-th key itself (keys are unique:
            // the position is part of them), so a plateau of exact ties is cut by position, lowest first, like every other path
            if (tid == 0) {
                uint32_t in_bins = 0;
                for (int b = 0; b < 256; ++b) in_bins += sh.hist[b];
                uint32_t cum = total - in_bins;            // records at or below klo
                int b = 0;
                for (; b < 255; ++b) { cum += sh.hist[b]; if (cum >= K1) break; }
                sh.bin = b;
            }
            __syncthreads();
            const unsigned long long mybin = (unsigned long long)sh.bin;
            const unsigned long long width = 1ull << sh_bits;
            const unsigned long long cut = klo + (mybin + 1ull) * width;          // (bin 255 also holds everything beyond it)
            if (mybin < 255ull && cut < khi) khi = cut;
            klo = klo + mybin * width;
            __syncthreads();
            continue;
        }
        __syncthreads();
        if (total >= K1 || total >= nfin || r >= r_all) break;
        klo = khi;                                // too few: grow towards the expected count (at most 2x per step), and at least to the box
        float rn = r * fminf(2.0f, fmaxf(1.26f, cbrtf(1.5f * (float)K1 / (float)max(total, 1u))));
        if (r < r_box) rn = fmaxf(rn, r_box);
        r = fminf(rn, r_all);
        khi = ((unsigned long long)__float_as_uint(r * r) << 32) | 0xFFFFFFFFull;
    }
    total = min(total, (uint32_t)CAPB);            // (cannot bind: the loop ends with K1 <= total <= CAPB, or with the whole cloud)
    // bitonic sort of the first n2 = 2^m >= total entries (padding: all ones)
    uint32_t n2 = 2 * kCoopThreads;
    while (n2 < total) n2 <<= 1;
    for (uint32_t i = total + tid; i < n2; i += kCoopThreads) sh.buf[i] = ~0ull;
    __syncthreads();
    for (uint32_t kk = 2; kk <= n2; kk <<= 1) {
        for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
            for (uint32_t t = (uint32_t)tid; t < (n2 >> 1); t += kCoopThreads) {
                const uint32_t i = 2 * t - (t & (jj - 1));          // the lower index of pair t at distance jj
                const uint32_t l = i + jj;
                const unsigned long long a = sh.buf[i], b = sh.buf[l];
                const bool up = (i & kk) == 0;
                if ((a > b) == up) { sh.buf[i] = b; sh.buf[l] = a; }
            }
            __syncthreads();
        }
    }
    return total;
}

// Radius mode beyond the register list (k_neighbors > 128 with a radius; the reference has no cap on either, normals.rs:17-26,
// :141-146): the block enumerates the ball of `radius` clipped to the grid -- rows dealt to the threads, closed-form x windows,
// like coop_nearest -- and folds count + first and second moments (about the query, f64) of every record within it except the
// query: mom[0] = count, mom[1..3] = sum d, mom[4..9] = sum d d^T (xx xy xz yy yz zz).  Fixed order (a thread's rows in ascending
// order, wave shuffle tree, waves in order): deterministic.  A set of >= k members IS the neighbourhood (order-free f64 moments +
// closed-form eigenvector, exactly what the lane-per-point kernel does with a radius set too large for its list); fewer: the k-NN
// path (normals.rs:315-323).
__device__ __forceinline__ void coop_radius_moments(const GridView &gv, float qx, float qy, float qz, uint32_t p, float radius, double (*red)[10],
                                                    double (&mom)[10]) {
    const GridGeom &g = gv.g;
    const int tid = threadIdx.x;
    const float r2 = radius * radius;
    double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float ry = radius * 1.0001f + 4e-3f * g.h;
    const int y0 = cell_coord(fminf(fmaxf(qy - ry, g.miny), g.maxy), g.miny, g.inv_h, g.gy), y1 = cell_coord(fminf(fmaxf(qy + ry, g.miny), g.maxy), g.miny, g.inv_h, g.gy);
    const int z0 = cell_coord(fminf(fmaxf(qz - ry, g.minz), g.maxz), g.minz, g.inv_h, g.gz), z1 = cell_coord(fminf(fmaxf(qz + ry, g.minz), g.maxz), g.minz, g.inv_h, g.gz);
    const int ny = y1 - y0 + 1;
    const uint32_t nrows = (uint32_t)ny * (uint32_t)(z1 - z0 + 1);
    for (uint32_t ri = (uint32_t)tid; ri < nrows; ri += kCoopThreads) {
        const int zz = z0 + (int)(ri / (uint32_t)ny), yy = y0 + (int)(ri % (uint32_t)ny);
        const float gy = g.clamped ? axis_gap_n<true>(qy, g.miny, g.h, yy, g.gy - 1) : axis_gap_n<false>(qy, g.miny, g.h, yy, g.gy - 1);
        const float gz = g.clamped ? axis_gap_n<true>(qz, g.minz, g.h, zz, g.gz - 1) : axis_gap_n<false>(qz, g.minz, g.h, zz, g.gz - 1);
        const float rg = gy * gy + gz * gz;
        if (rg > r2) continue;
        const float rx = sqrtf(fmaxf(r2 - rg, 0.0f)) * 1.0001f + 4e-3f * g.h;
        const int xa = (int)fminf(fmaxf((qx - rx - g.minx) * g.inv_h, 0.0f), (float)(g.gx - 1));
        const int xb = (int)fmaxf(fminf((qx + rx - g.minx) * g.inv_h, (float)(g.gx - 1)), 0.0f);
        if (xa > xb) continue;
        const uint32_t row = ((uint32_t)zz * g.gy + yy) * g.gx;
        const uint32_t s = gv.cell_start[row + xa], e = gv.cell_start[row + xb + 1];
        for (uint32_t j = s; j < e; ++j) {
            const float4 c = gv.pts[j];
            const float v = d2_nc(c.x, c.y, c.z, qx, qy, qz);
            if (j == p || !(v <= r2)) continue;                      // nearest_neighbor.rs:254-298: d2 <= r2; the query itself is dropped
            const double dx = (double)c.x - (double)qx, dy = (double)c.y - (double)qy, dz = (double)c.z - (double)qz;
            a[0] += 1.0;
            a[1] += dx; a[2] += dy; a[3] += dz;
            a[4] = fma(dx, dx, a[4]); a[5] = fma(dx, dy, a[5]); a[6] = fma(dx, dz, a[6]);
            a[7] = fma(dy, dy, a[7]); a[8] = fma(dy, dz, a[8]); a[9] = fma(dz, dz, a[9]);
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double v = a[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        a[i] = v;
    }
    __syncthreads();                                  // (red may still be read by the previous point's consumers)
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) red[tid >> 6][i] = a[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kCoopThreads / 64; ++w) v += red[w][i];
        mom[i] = v;
    }
}

template <int CAPB>
__global__ void __launch_bounds__(kCoopThreads) normals_coop_kernel(GridView gv, NormalParams prm, float *__restrict__ out6, uint32_t *__restrict__ hard) {
    __shared__ CoopShared<CAPB> sh;
    __shared__ float nbx[CAPB / 2], nby[CAPB / 2], nbz[CAPB / 2];
    __shared__ int self_s;
    __shared__ double rmom[kCoopThreads / 64][10];
    const GridGeom &g = gv.g;
    const int tid = threadIdx.x;
    const uint32_t count = hard ? hard[0] : prm.p_end - prm.p_begin;
    const uint32_t nfin = gv.cell_start[g.ncell];
    for (uint32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const uint32_t p = hard ? hard[4u + idx] : prm.p_begin + idx;
        const float4 q = gv.pts[p];
        const uint32_t orig = __float_as_uint(q.w);
        float nrm_x = 0.0f, nrm_y = 0.0f, nrm_z = 1.0f;
        float vor_w = 0.0f;
        float px = q.x, py = q.y, pz = q.z;
        if (p >= nfin) {            // a non-finite point: the default normal (see normals_point)
            if (tid == 0) { px = prm.xyz[3 * (size_t)orig]; py = prm.xyz[3 * (size_t)orig + 1]; pz = prm.xyz[3 * (size_t)orig + 2]; }
        } else {
            bool by_radius = false;
            if (prm.has_radius && !hard) {
                double mom[10];
                coop_radius_moments(gv, q.x, q.y, q.z, p, prm.radius, rmom, mom);
                by_radius = mom[0] >= (double)prm.k;                 // normals.rs:315: fewer than k members -> the k-NN fallback
                if (by_radius && tid == 0) {
                    const double nn = mom[0] + 1.0;                  // + the query itself (normals.rs:338-340), at offset 0
                    const double mx = mom[1] / nn, my = mom[2] / nn, mz = mom[3] / nn;
                    double ex, ey, ez;
                    smallest_eigvec_sym3(mom[4] / nn - mx * mx, mom[5] / nn - mx * my, mom[6] / nn - mx * mz, mom[7] / nn - my * my,
                                         mom[8] / nn - my * mz, mom[9] / nn - mz * mz, ex, ey, ez);
                    const float vx = (float)ex, vy = (float)ey, vz = (float)ez;
                    const float mag = sqrtf(vx * vx + vy * vy + vz * vz);
                    if (mag > 1e-6f) { nrm_x = vx / mag; nrm_y = vy / mag; nrm_z = vz / mag; }
                    if (prm.orient) {
                        const float tx = prm.vx - q.x, ty = prm.vy - q.y, tz = prm.vz - q.z;
                        const float tn = sqrtf(tx * tx + ty * ty + tz * tz);
                        const float dp = nrm_x * (tx / tn) + nrm_y * (ty / tn) + nrm_z * (tz / tn);
                        if (dp < 0.0f) { nrm_x = -nrm_x; nrm_y = -nrm_y; nrm_z = -nrm_z; }
                    }
                }
            }
            if (!by_radius) {
            const uint32_t K1 = min(prm.k + 1u, nfin);
            if (tid == 0) self_s = -1;
            const uint32_t total = coop_nearest<CAPB>(gv, q.x, q.y, q.z, K1, nfin, sh);
            const uint32_t cnt = min(K1, total);        // the k + 1 nearest (fewer: the whole cloud)
            // their coordinates, parked in LDS by all lanes
            for (uint32_t r2 = (uint32_t)tid; r2 < cnt; r2 += kCoopThreads) {
                const uint32_t j = (uint32_t)sh.buf[r2];
                const float4 c = gv.pts[j];
                nbx[r2] = c.x; nby[r2] = c.y; nbz[r2] = c.z;
                if (j == p) self_s = (int)r2;
            }
            __syncthreads();
            const int self_r = self_s;
            vor_w = 0.25f * 0.9999f * (cnt >= 2 ? __uint_as_float((uint32_t)(sh.buf[1] >> 32)) : INFINITY);
            if (tid == 0) {
                // normals.rs:147-153: drop self from the k + 1 list (or the last entry when self is not in it); self appended last
                const int drop_r = (self_r >= 0) ? self_r : (int)cnt - 1;
                const uint32_t npts = cnt;
                if (npts >= 3) {
                    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
                    for (uint32_t r2 = 0; r2 < cnt; ++r2) {
                        if ((int)r2 == drop_r) continue;
                        sx += nbx[r2]; sy += nby[r2]; sz += nbz[r2];
                    }
                    sx += q.x; sy += q.y; sz += q.z;
                    const float nf = (float)npts;
                    const float mx = sx / nf, my = sy / nf, mz = sz / nf;
                    float cxx = 0.0f, cxy = 0.0f, cxz = 0.0f, cyy = 0.0f, cyz = 0.0f, czz = 0.0f;
                    for (uint32_t r2 = 0; r2 < cnt; ++r2) {
                        if ((int)r2 == drop_r) continue;
                        const float dx = nbx[r2] - mx, dy = nby[r2] - my, dz = nbz[r2] - mz;
                        cxx += dx * dx; cxy += dx * dy; cxz += dx * dz; cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
                    }
                    {
                        const float dx = q.x - mx, dy = q.y - my, dz = q.z - mz;
                        cxx += dx * dx; cxy += dx * dy; cxz += dx * dz; cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
                    }
                    cxx /= nf; cxy /= nf; cxz /= nf; cyy /= nf; cyz /= nf; czz /= nf;
                    float e0, e1, e2, x0, y0, z0, x1, y1, z1, x2, y2, z2;
                    sym_eigen3_f32(cxx, cxy, cxz, cyy, cyz, czz, e0, e1, e2, x0, y0, z0, x1, y1, z1, x2, y2, z2);
                    float vx = x0, vy = y0, vz = z0, emin = e0;
                    if (e1 < emin) { emin = e1; vx = x1; vy = y1; vz = z1; }
                    if (e2 < emin) { vx = x2; vy = y2; vz = z2; }
                    const float mag = sqrtf(vx * vx + vy * vy + vz * vz);
                    if (mag > 1e-6f) { nrm_x = vx / mag; nrm_y = vy / mag; nrm_z = vz / mag; }
                }
                if (prm.orient) {
                    const float tx = prm.vx - q.x, ty = prm.vy - q.y, tz = prm.vz - q.z;
                    const float tn = sqrtf(tx * tx + ty * ty + tz * tz);
                    const float ux = tx / tn, uy = ty / tn, uz = tz / tn;
                    const float dp = nrm_x * ux + nrm_y * uy + nrm_z * uz;
                    if (dp < 0.0f) { nrm_x = -nrm_x; nrm_y = -nrm_y; nrm_z = -nrm_z; }
                }
            }
            }   // !by_radius
        }
        if (tid == 0) {
            if (prm.sorted_nrm) prm.sorted_nrm[p] = make_float4(nrm_x, nrm_y, nrm_z, 0.0f);
            if (prm.vor_out) prm.vor_out[p] = make_float4(q.x, q.y, q.z, vor_w);
            if (out6) {
                float *o = out6 + 6 * (size_t)(prm.slice_out ? p - prm.p_begin : orig);
                o[0] = px; o[1] = py; o[2] = pz; o[3] = nrm_x; o[4] = nrm_y; o[5] = nrm_z;
            }
        }
        __syncthreads();
    }
    // the list is left empty for the next normals launch by the LAST block to leave (every block has read the count by then): no
    // memset launch per call
    if (hard && tid == 0 && atomicAdd(&hard[1], 1u) == gridDim.x - 1u) { hard[0] = 0u; hard[1] = 0u; hard[2] = 0u; hard[3] = 0u; }
}

// NearestNeighborSearch::find_k_nearest beyond the register list's 129 entries (k up to 2048): a block per query, the same
// selection; output like knn_kernel: (original index, sqrt(d2)) ascending, count = the entries within radius_sq
template <int CAPB>
__global__ void __launch_bounds__(kCoopThreads) knn_coop_kernel(GridView gv, const float *__restrict__ queries, uint32_t nq, uint32_t k,
                                                                uint32_t *__restrict__ out_idx, float *__restrict__ out_dist,
                                                                uint32_t *__restrict__ out_count, float radius_sq) {
    __shared__ CoopShared<CAPB> sh;
    __shared__ uint32_t within_s;
    const GridGeom &g = gv.g;
    const int tid = threadIdx.x;
    const uint32_t nfin = gv.cell_start[g.ncell];
    for (uint32_t t = blockIdx.x; t < nq; t += gridDim.x) {
        const float qx = queries[3 * (size_t)t], qy = queries[3 * (size_t)t + 1], qz = queries[3 * (size_t)t + 2];
        const uint32_t K1 = min(k, nfin);
        // a NaN / infinite query has no finite distance to anything: no neighbours (see knn_kernel)
        if (!(fabsf(qx) <= 3.0e38f && fabsf(qy) <= 3.0e38f && fabsf(qz) <= 3.0e38f) || K1 == 0) {
            if (tid == 0) out_count[t] = 0;
            continue;
        }
        if (tid == 0) within_s = 0;
        const uint32_t total = coop_nearest<CAPB>(gv, qx, qy, qz, K1, nfin, sh);
        const uint32_t cnt = min(K1, total);
        uint32_t within = 0;
        for (uint32_t r = (uint32_t)tid; r < cnt; r += kCoopThreads) {
            const unsigned long long key = sh.buf[r];
            const float v = __uint_as_float((uint32_t)(key >> 32));
            out_idx[(size_t)t * k + r] = __float_as_uint(gv.pts[(uint32_t)key].w);
            out_dist[(size_t)t * k + r] = sqrtf(v);                                   // nearest_neighbor.rs:249
            within += (v <= radius_sq) ? 1u : 0u;
        }
        if (within) atomicAdd(&within_s, within);
        __syncthreads();
        if (tid == 0) out_count[t] = within_s;
        __syncthreads();
    }
}

// XCD-aware block remap: hardware deals blocks round-robin over the 8 XCDs, so give each XCD
// one contiguous eighth of the cell-sorted array (its L2 then holds a contiguous slab + halo).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb) {
    const uint32_t per = (nb + 7) / 8;
#ifdef TC_NORMALS_REVERSE
    const uint32_t lb = (b & 7u) * per + (per - 1u - (b >> 3));          // (A/B: each XCD walks its slab from the far end)
#else
    const uint32_t lb = (b & 7u) * per + (b >> 3);
#endif
    return lb;
}

template <int L, int BLOCK, bool RADIUS, bool EXT, int CAP = 0>
__global__ void __launch_bounds__(BLOCK) normals_knn_pca_kernel(GridView gv, NormalParams prm, float *__restrict__ out6) {
    __shared__ uint32_t ldsA[L * BLOCK];
    __shared__ uint8_t ldsB[L * BLOCK];       // ranks as bytes: 88 instead of 136 B of LDS per lane -> 7 instead of 4 waves per SIMD
    const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t p = prm.p_begin + lb * BLOCK + threadIdx.x;
    if (p >= prm.p_end) return;
#ifdef TC_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    normals_point<L, BLOCK, RADIUS, EXT, CAP>(gv, prm, p, out6, ldsA + threadIdx.x, ldsB + threadIdx.x, ph, tl);
    if (threadIdx.x == 0 && prm.stamps) for (int i = 0; i < 8; ++i) prm.stamps[8 * (size_t)blockIdx.x + i] = ph[i];
#else
    normals_point<L, BLOCK, RADIUS, EXT, CAP>(gv, prm, p, out6, ldsA + threadIdx.x, ldsB + threadIdx.x);
#endif
}

// the tagged-key instantiation, held to the register-list path's six waves per SIMD (its decode step would otherwise keep all
// its gathers in flight at once: 191 VGPRs); five for the 19-entry list and above: at six it spills 21 registers (96 B of scratch
// per lane = 84 MB of extra write traffic at 1 M points) for the same time (321 vs 325 us)
#ifndef TC_TAG_WAVES
#define TC_TAG_WAVES 6
#endif
// One wave per block for the flattened tagged kernels: a block's LDS and wave slots are held until its slowest wave is done, and a
// wave's time spreads over p10 - p90 = 106 k - 179 k cycles: 320 -> 293 us at 1 M uniform points / k = 16 (128 threads: 297; no effect
// at k = 10, and none on the register-list kernels: 438 -> 433)
#ifndef TC_TAG_BLOCK
#define TC_TAG_BLOCK 64
#endif
template <int L, int BLOCK, bool EXT, int CAP>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(L >= 19 ? TC_TAG_WAVES - 1 : TC_TAG_WAVES, L >= 19 ? TC_TAG_WAVES - 1 : TC_TAG_WAVES))) normals_tagged_kernel(GridView gv, NormalParams prm, float *__restrict__ out6) {
    __shared__ uint32_t ldsA[(L > kFlatSpanWords ? L : kFlatSpanWords) * BLOCK];        // (the flattened walk parks the spans of a group here, two words each)
    __shared__ uint8_t ldsB[L * BLOCK];
    const uint32_t lb = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t p = prm.p_begin + lb * BLOCK + threadIdx.x;
    if (p >= prm.p_end) return;
#ifdef TC_PHASE_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
    normals_point<L, BLOCK, false, EXT, CAP>(gv, prm, p, out6, ldsA + threadIdx.x, ldsB + threadIdx.x, ph, tl);
    if (threadIdx.x == 0 && prm.stamps) for (int i = 0; i < 8; ++i) prm.stamps[8 * (size_t)blockIdx.x + i] = ph[i];
#else
    normals_point<L, BLOCK, false, EXT, CAP>(gv, prm, p, out6, ldsA + threadIdx.x, ldsB + threadIdx.x);
#endif
}

template <int L, int BLOCK, bool RADIUS = false, int CAP = 0>
static void launch_variant(hipStream_t st, const GridView &gv, const NormalParams &prm, float *out6, tc_context *ctx) {
    const uint32_t n = prm.p_end - prm.p_begin;
    if (n == 0) return;
    uint32_t nb = (n + BLOCK - 1) / BLOCK;
    nb = (nb + 7) / 8 * 8;   // xcd_remap needs a multiple of 8
    ProfScope ps(ctx, "normals_knn_pca");
    // two instantiations: with a clamped box the boundary cells are open on the outer side (costs 4 % on the gap tests)
    if constexpr (CAP < 0) {
        if (gv.g.clamped) hipLaunchKernelGGL((normals_tagged_kernel<L, BLOCK, true, CAP>), dim3(nb), dim3(BLOCK), 0, st, gv, prm, out6);
        else hipLaunchKernelGGL((normals_tagged_kernel<L, BLOCK, false, CAP>), dim3(nb), dim3(BLOCK), 0, st, gv, prm, out6);
    } else {
        if (gv.g.clamped) hipLaunchKernelGGL((normals_knn_pca_kernel<L, BLOCK, RADIUS, true, CAP>), dim3(nb), dim3(BLOCK), 0, st, gv, prm, out6);
        else hipLaunchKernelGGL((normals_knn_pca_kernel<L, BLOCK, RADIUS, false, CAP>), dim3(nb), dim3(BLOCK), 0, st, gv, prm, out6);
    }
}

// ---- batch k-NN export -----------------------------------------------------
// NearestNeighborSearch::find_k_nearest (nearest_neighbor.rs:177-251, trait core/traits.rs:6-12;
// gpu_find_k_nearest_batch threecrate-gpu/src/nearest_neighbor.rs:345-355): for every query the k
// nearest cloud points, ascending, as (original index, sqrt(d2)).  Same machinery as the normals
// kernel: sorted register list for the k-th distance, ball-pruned ring continuation (queries may lie
// outside the grid: |p - q|^2 >= |p - clamp(q)|^2 + |q - clamp(q)|^2), LDS position lists, ranking.
template <int L, int BLOCK, bool EXT>
__global__ void __launch_bounds__(BLOCK) knn_kernel(GridView gv, const float *__restrict__ queries, uint32_t nq, uint32_t k,
                                                    uint32_t *__restrict__ out_idx, float *__restrict__ out_dist,
                                                    uint32_t *__restrict__ out_count, float radius_sq) {
    __shared__ uint32_t ldsA_[L * BLOCK];
    __shared__ uint8_t ldsB_[L * BLOCK];
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nq) return;
    uint32_t *ldsA = ldsA_ + threadIdx.x;
    uint8_t *ldsB = ldsB_ + threadIdx.x;
    const GridGeom &g = gv.g;
    float4 q;
    q.x = queries[3 * (size_t)t]; q.y = queries[3 * (size_t)t + 1]; q.z = queries[3 * (size_t)t + 2]; q.w = 0.0f;
    // a NaN / infinite query has no finite distance to anything: no neighbours (the reference's kd-tree returns whatever nodes
    // its NaN comparisons visit first, with NaN distances)
    if (!(fabsf(q.x) <= 3.0e38f && fabsf(q.y) <= 3.0e38f && fabsf(q.z) <= 3.0e38f)) { out_count[t] = 0; return; }
    const float qx = fminf(fmaxf(q.x, g.minx), g.maxx), qy = fminf(fmaxf(q.y, g.miny), g.maxy), qz = fminf(fmaxf(q.z, g.minz), g.maxz);
    const int cx = cell_coord(qx, g.minx, g.inv_h, g.gx), cy = cell_coord(qy, g.miny, g.inv_h, g.gy), cz = cell_coord(qz, g.minz, g.inv_h, g.gz);
    const float fx = (qx - g.minx) * g.inv_h - (float)cx, fy = (qy - g.miny) * g.inv_h - (float)cy, fz = (qz - g.minz) * g.inv_h - (float)cz;
    const float mf = fmaxf(fminf(fminf(fminf(fx, 1.0f - fx), fminf(fy, 1.0f - fy)), fminf(fz, 1.0f - fz)), 0.0f);
    const float ex = q.x - qx, ey = q.y - qy, ez = q.z - qz;
    // |p - q|^2 >= |p - q'|^2 + |q - q'|^2 needs every record inside the box: not so when the box is clamped
    const float out2 = EXT ? 0.0f : (ex * ex + ey * ey + ez * ez) * 0.9999f;
    const uint32_t K1 = min(k, gv.cell_start[g.ncell]);      // the finite points
    if (K1 == 0) { out_count[t] = 0; return; }
    float d[L];
#pragma unroll
    for (int i = 0; i < L; ++i) d[i] = INFINITY;
    auto visit1 = [&](uint32_t, const float4 &c) { list_insert<L>(d, d2_nc(c.x, c.y, c.z, q.x, q.y, q.z)); };
    int R = 1;
    float tau = INFINITY;
    scan_block(gv, cx, cy, cz, R, visit1);
    for (;;) {
        tau = d[0];
#pragma unroll
        for (int i = 1; i < L; ++i) tau = ((uint32_t)i == K1 - 1) ? d[i] : tau;
        const bool covers = (cx - R <= 0) && (cx + R >= g.gx - 1) && (cy - R <= 0) && (cy + R >= g.gy - 1) &&
                            (cz - R <= 0) && (cz + R >= g.gz - 1);
        const float bound = ((float)R + mf - 2e-3f) * g.h;
        if (covers || tau <= bound * bound + out2) break;
        const int Rin = R;                                       // see normals_point
        if (tau == INFINITY) R += max(1, R / 2);
        else R = max(R + 1, (int)fminf(ceilf(sqrtf(fmaxf(tau - out2, 0.0f)) * g.inv_h - mf + 0.01f), 1.0e9f));
        const bool growing = tau == INFINITY || R > Rin + 1;
        float live_lim = tau;
        const bool touched = scan_pruned<EXT, true>(gv, q, cx, cy, cz, Rin, R, live_lim, [&](uint32_t j, const float4 &c) {
            visit1(j, c);
            if (growing) live_lim = d[L - 1];       // bounds the k-th entry (static index: see normals_point)
        }, &live_lim);
        if (!touched) {
            tau = d[0];
#pragma unroll
            for (int i = 1; i < L; ++i) tau = ((uint32_t)i == K1 - 1) ? d[i] : tau;
            break;
        }
    }
    uint32_t n_lt = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) n_lt += (d[i] < tau) ? 1u : 0u;
    const uint32_t quota = K1 - min(n_lt, K1);
    uint32_t cnt = 0, ties = 0;
    scan_pruned<EXT>(gv, q, cx, cy, cz, -1, R, tau, [&](uint32_t j, const float4 &c) {
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        bool take = v < tau;
        if (!take && v == tau && ties < quota) { take = true; ++ties; }
        if (take && cnt < K1) { ldsA[cnt * BLOCK] = j; ++cnt; }
    });
    unsigned long long taken_lo = 0ull, taken_hi = 0ull, taken_x = 0ull;
    auto is_taken = [&](uint32_t r) { return r < 64 ? ((taken_lo >> r) & 1ull) : r < 128 ? ((taken_hi >> (r - 64)) & 1ull) : ((taken_x >> (r - 128)) & 1ull); };
    for (uint32_t e = 0; e < cnt; ++e) {
        const uint32_t j = ldsA[e * BLOCK];
        const float4 c = gv.pts[j];
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) r += (d[i] < v) ? 1u : 0u;
        while (is_taken(r)) ++r;
        if (r < 64) taken_lo |= 1ull << r; else if (r < 128) taken_hi |= 1ull << (r - 64); else taken_x |= 1ull << (r - 128);
        ldsB[r * BLOCK] = (uint8_t)e;
    }
    uint32_t within = 0;          // radius search: the entries with d2 <= radius^2 (nearest_neighbor.rs:271), a prefix
    for (uint32_t r = 0; r < cnt; ++r) {
        const float4 c = gv.pts[ldsA[(uint32_t)ldsB[r * BLOCK] * BLOCK]];
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        out_idx[(size_t)t * k + r] = __float_as_uint(c.w);
        out_dist[(size_t)t * k + r] = sqrtf(v);                                       // nearest_neighbor.rs:249
        within += (v <= radius_sq) ? 1u : 0u;
    }
    out_count[t] = within;
}

// ---- unbounded radius search: NearestNeighborSearch::find_radius_neighbors (nearest_neighbor.rs:254-298) ----------
// every cloud point with d2 <= radius^2, two launches: count per query, then fill at the caller's offsets (grid scan
// order; the callers sort by distance like the reference's final sort_by).
template <bool EXT, bool FILL>
__global__ void __launch_bounds__(128) radius_all_kernel(GridView gv, const float *__restrict__ queries, uint32_t nq, float radius,
                                                        uint32_t *__restrict__ counts, const unsigned long long *__restrict__ offsets,
                                                        uint32_t *__restrict__ out_idx, float *__restrict__ out_dist) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq) return;
    const GridGeom &g = gv.g;
    float4 q;
    q.x = queries[3 * (size_t)t]; q.y = queries[3 * (size_t)t + 1]; q.z = queries[3 * (size_t)t + 2]; q.w = 0.0f;
    if (!(fabsf(q.x) <= 3.0e38f && fabsf(q.y) <= 3.0e38f && fabsf(q.z) <= 3.0e38f)) { if (!FILL) counts[t] = 0; return; }
    const float qx = fminf(fmaxf(q.x, g.minx), g.maxx), qy = fminf(fmaxf(q.y, g.miny), g.maxy), qz = fminf(fmaxf(q.z, g.minz), g.maxz);
    const int cx = cell_coord(qx, g.minx, g.inv_h, g.gx), cy = cell_coord(qy, g.miny, g.inv_h, g.gy), cz = cell_coord(qz, g.minz, g.inv_h, g.gz);
    const float r2 = radius * radius;                                                 // nearest_neighbor.rs:259
    // cells further than this from the query's (clamped) cell cannot hold a point of the ball
    const int R = (int)fminf(ceilf(radius * g.inv_h) + 1.0f, (float)max(g.gx, max(g.gy, g.gz)));
    const unsigned long long base = FILL ? offsets[t] : 0ull;
    uint32_t cnt = 0;
    scan_pruned<EXT>(gv, q, cx, cy, cz, -1, R, r2, [&](uint32_t, const float4 &c) {
        const float v = d2_nc(c.x, c.y, c.z, q.x, q.y, q.z);
        if (v <= r2) {                                                                 // :271
            if (FILL) { out_idx[base + cnt] = __float_as_uint(c.w); out_dist[base + cnt] = sqrtf(v); }
            ++cnt;
        }
    });
    if (!FILL) counts[t] = cnt;
}

tc_status launch_radius_all(tc_context *ctx, const DeviceIndex &ix, const float *d_queries, size_t nq, float radius, uint32_t *d_counts,
                            const unsigned long long *d_offsets, uint32_t *d_idx, float *d_dist) {
    const GridView gv = view_of(ix);
    ProfScope ps(ctx, d_offsets ? "radius_fill" : "radius_count");
    const dim3 grid((unsigned)((nq + 127) / 128)), block(128);
    hipStream_t st = ctx->stream;
    if (!d_offsets) {
        if (gv.g.clamped) hipLaunchKernelGGL((radius_all_kernel<true, false>), grid, block, 0, st, gv, d_queries, (uint32_t)nq, radius, d_counts, nullptr, nullptr, nullptr);
        else hipLaunchKernelGGL((radius_all_kernel<false, false>), grid, block, 0, st, gv, d_queries, (uint32_t)nq, radius, d_counts, nullptr, nullptr, nullptr);
    } else {
        if (gv.g.clamped) hipLaunchKernelGGL((radius_all_kernel<true, true>), grid, block, 0, st, gv, d_queries, (uint32_t)nq, radius, nullptr, d_offsets, d_idx, d_dist);
        else hipLaunchKernelGGL((radius_all_kernel<false, true>), grid, block, 0, st, gv, d_queries, (uint32_t)nq, radius, nullptr, d_offsets, d_idx, d_dist);
    }
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

tc_status launch_knn(tc_context *ctx, const DeviceIndex &ix, const float *d_queries, size_t nq, size_t k,
                     uint32_t *d_idx, float *d_dist, uint32_t *d_count, float radius_sq) {
    if (k > 2048) return fail(ctx, TC_UNSUPPORTED, "k > 2048 is not supported by the HIP k-NN export");
    const GridView gv = view_of(ix);
    ProfScope ps(ctx, "knn_batch");
    hipStream_t st = ctx->stream;
    if (k > 129) {          // beyond the register list: a block per query (knn_coop_kernel)
        const dim3 grid((unsigned)std::min<size_t>(nq, 1u << 16)), block(kCoopThreads);
        if (k <= 256) hipLaunchKernelGGL(knn_coop_kernel<512>, grid, block, 0, st, gv, d_queries, (uint32_t)nq, (uint32_t)k, d_idx, d_dist, d_count, radius_sq);
        else hipLaunchKernelGGL(knn_coop_kernel<4096>, grid, block, 0, st, gv, d_queries, (uint32_t)nq, (uint32_t)k, d_idx, d_dist, d_count, radius_sq);
        TC_HIP_TRY(ctx, hipGetLastError());
        return TC_OK;
    }
#define TC_KNN(LL, BB)                                                                                                          \
    do {                                                                                                                        \
        if (gv.g.clamped) hipLaunchKernelGGL((knn_kernel<LL, BB, true>), dim3((unsigned)((nq + BB - 1) / BB)), dim3(BB), 0, st, gv, \
                                             d_queries, (uint32_t)nq, (uint32_t)k, d_idx, d_dist, d_count, radius_sq);          \
        else hipLaunchKernelGGL((knn_kernel<LL, BB, false>), dim3((unsigned)((nq + BB - 1) / BB)), dim3(BB), 0, st, gv, d_queries, \
                                (uint32_t)nq, (uint32_t)k, d_idx, d_dist, d_count, radius_sq);                                  \
    } while (0)
    if (k <= 9) TC_KNN(9, 256);
    else if (k <= 17) TC_KNN(17, 256);
    else if (k <= 33) TC_KNN(33, 128);
    else if (k <= 65) TC_KNN(65, 64);
    else TC_KNN(129, 64);
#undef TC_KNN
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

// out[6 * orig(p) ..] = sorted[6 * p ..]: the gathered slices of a sharded run back into input order
__global__ void __launch_bounds__(256) normals_unsort_kernel(const float4 *__restrict__ pts, uint32_t n, const float *__restrict__ sorted6,
                                                            float *__restrict__ out6) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t orig = __float_as_uint(pts[p].w);
    const float *r = sorted6 + 6 * (size_t)p;
    float *o = out6 + 6 * (size_t)orig;
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = r[c];
}

tc_status launch_normals_unsort(tc_context *ctx, const DeviceIndex &ix, const float *d_sorted6, float *d_out6) {
    const uint32_t n = ix.geom.n;
    hipLaunchKernelGGL(normals_unsort_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float4 *)ix.pts.p, n, d_sorted6, d_out6);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

tc_status launch_normals(tc_context *ctx, const DeviceIndex &ix, const float *d_xyz, const tc_normal_config &cfg, const float vp[3],
                         float *d_out6, size_t p_begin, size_t p_end, bool slice_out, float4 *d_sorted_nrm, float4 *d_vor) {
    const bool radius_mode = cfg.has_radius && cfg.radius > 0.0f;
    // k_neighbors > 128: the wave-per-point kernel (the k + 1 nearest in an LDS buffer instead of a register list), up to 2047
    const bool big_k = cfg.k_neighbors + 1 > 129;
    // (k_neighbors > 128 together with a radius: the wave-per-point kernel folds the radius ball's moments itself -- coop_radius_moments)
    if (cfg.k_neighbors + 1 > 2048) return fail(ctx, TC_UNSUPPORTED, "k_neighbors > 2047 is not supported by the HIP backend");
    NormalParams prm;
    prm.hard_list = nullptr;
    prm.tag_policy = 0;
    prm.k = (uint32_t)cfg.k_neighbors;
    prm.orient = cfg.consistent_orientation ? 1 : 0;
    prm.vx = vp[0]; prm.vy = vp[1]; prm.vz = vp[2];
    prm.R0 = 2;
    prm.has_radius = (cfg.has_radius && cfg.radius > 0.0f) ? 1 : 0;
    prm.radius = prm.has_radius ? cfg.radius : 0.0f;
    prm.xyz = d_xyz;
    prm.sorted_nrm = d_sorted_nrm;
    prm.vor_out = d_vor;
    prm.p_begin = (uint32_t)p_begin;
    prm.p_end = (uint32_t)std::min<size_t>(p_end, ix.geom.n);
    prm.slice_out = slice_out ? 1 : 0;
#ifdef TC_PHASE_STAMPS
    prm.stamps = nullptr;
    const size_t stamp_blocks = ((prm.p_end - prm.p_begin) / 64 + 16);
    if (debug_flags() & 1024) {
        if (tc_status s = ensure(ctx, ctx->dbg_times, 8 * stamp_blocks * sizeof(unsigned long long))) return s;
        (void)hipMemsetAsync(ctx->dbg_times.p, 0, 8 * stamp_blocks * sizeof(unsigned long long), ctx->stream);
        prm.stamps = (unsigned long long *)ctx->dbg_times.p;
    }
    struct StampDump {
        tc_context *ctx; size_t nb; bool on;
        ~StampDump() {
            if (!on) return;
            (void)hipStreamSynchronize(ctx->stream);
            std::vector<unsigned long long> h(8 * nb);
            (void)hipMemcpy(h.data(), ctx->dbg_times.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double m[8] = {0}; size_t cnt = 0;
            for (size_t b = 0; b < nb; ++b) { if (!h[8 * b + 1]) continue; ++cnt; for (int i = 0; i < 8; ++i) m[i] += (double)h[8 * b + i]; }
            for (int i = 0; i < 8; ++i) m[i] /= std::max<size_t>(cnt, 1);
            {
                // spread of a block's wave-0 time (all phases) and where the long ones sit in launch order
                std::vector<double> tot;
                for (size_t b = 0; b < nb; ++b) { if (!h[8 * b + 1]) continue; double t = 0; for (int i = 0; i < 8; ++i) t += (double)h[8 * b + i]; tot.push_back(t); }
                if (!tot.empty()) {
                    std::vector<double> srt = tot;
                    std::sort(srt.begin(), srt.end());
                    auto q = [&](double f) { return srt[std::min(srt.size() - 1, (size_t)(f * (double)srt.size()))]; };
                    const size_t tenth = std::max<size_t>(tot.size() / 10, 1);
                    fprintf(stderr, "[tc] normals wave 0 total ticks per block: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | mean by tenth of the launch order:", q(0.1), q(0.5), q(0.9), q(0.99), srt.back());
                    for (size_t d0 = 0; d0 + tenth <= tot.size(); d0 += tenth) { double a = 0; for (size_t i = d0; i < d0 + tenth; ++i) a += tot[i]; fprintf(stderr, " %.0f", a / (double)tenth); }
                    fprintf(stderr, "\n");
                }
            }
            fprintf(stderr, "[tc] normals wave 0 phases, mean s_memtime ticks over %zu blocks: setup %.0f  block scan + list (tagged path: exactness rule + ring-3 continuation) %.0f  continuation (tagged: row logic of the three groups) %.0f  collect (tagged: the flattened walks) %.0f  rank (tagged: decode + certificate) %.0f  centroid + covariance %.0f  eigen + orient %.0f  store %.0f\n",
                    cnt, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        }
    } stamp_dump{ctx, stamp_blocks, prm.stamps != nullptr};
#endif
    const GridView gv = view_of(ix);
    const uint32_t K1 = prm.k + 1;
    if (big_k) {
        const uint32_t npts = prm.p_end - prm.p_begin;
        if (npts == 0) return TC_OK;
        ProfScope ps(ctx, "normals_coop");
        const dim3 grid(std::min<uint32_t>(npts, 1u << 16)), block(kCoopThreads);
        if (K1 <= 256) hipLaunchKernelGGL(normals_coop_kernel<512>, grid, block, 0, ctx->stream, gv, prm, d_out6, (uint32_t *)nullptr);
        else hipLaunchKernelGGL(normals_coop_kernel<4096>, grid, block, 0, ctx->stream, gv, prm, d_out6, (uint32_t *)nullptr);
        TC_HIP_TRY(ctx, hipGetLastError());
        return TC_OK;
    }
    // hard points (isolated: their search would walk the grid through one lane) are listed by the main launch and served by the
    // wave-per-point kernel behind it (k-NN mode)
    if (!radius_mode && prm.p_end > prm.p_begin) {
        const void *before = ctx->normals_hard.p;
        if (tc_status s = ensure(ctx, ctx->normals_hard, ((size_t)(prm.p_end - prm.p_begin) + 4) * sizeof(uint32_t))) return s;
        prm.hard_list = (uint32_t *)ctx->normals_hard.p;
        // (count and exit ticket are zeroed when the buffer is new or the last serving launch is not known to have gone through;
        // afterwards the serving kernel leaves them zero)
        if (ctx->normals_hard.p != before || !ctx->normals_hard_clean) {
            TC_HIP_TRY(ctx, hipMemsetAsync(prm.hard_list, 0, 4 * sizeof(uint32_t), ctx->stream));
            ctx->normals_hard_clean = true;
        }
    }
    struct HardPass {
        tc_context *ctx; const GridView &gv; NormalParams &prm; float *out6;
        ~HardPass() {
            if (!prm.hard_list) return;
            uint32_t *hl = prm.hard_list;
            NormalParams p2 = prm;
            p2.hard_list = nullptr;
            ProfScope ps(ctx, "normals_coop");
            ctx->normals_hard_clean = false;       // until the launch that resets the header is known to be enqueued
            hipLaunchKernelGGL(normals_coop_kernel<512>, dim3(256), dim3(kCoopThreads), 0, ctx->stream, gv, p2, out6, hl);
            if (hipGetLastError() == hipSuccess) ctx->normals_hard_clean = true;
        }
    } hard_pass{ctx, gv, prm, d_out6};
#ifdef TC_NSTATS
    {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nstats), z, sizeof z);
    }
    struct StatDump {
        hipStream_t st;
        ~StatDump() {
            (void)hipStreamSynchronize(st);
            unsigned long long h[16];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_nstats), sizeof h);
            fprintf(stderr, "[tc] normals tagged path: lanes %llu served %llu | beyond ring 3 / oversized span %llu | a check failed %llu | waves %llu with a fallback lane %llu\n",
                    h[0] + h[2], h[1], h[2], h[3], h[6], h[7]);
            const double lanes = (double)std::max<unsigned long long>(h[0] + h[2], 1ull), waves = (double)std::max<unsigned long long>(h[6], 1ull);
            fprintf(stderr, "[tc] normals flattened walk, steps of %d records: a lane needs %.1f + %.1f + %.1f = %.1f per point (groups 0 / 1 / 2), its wave takes %.1f + %.1f + %.1f = %.1f: lock-step ratio %.2f\n",
                    TC_FLAT_W, h[8] / lanes, h[9] / lanes, h[10] / lanes, (h[8] + h[9] + h[10]) / lanes, h[11] / waves, h[12] / waves, h[13] / waves,
                    (h[11] + h[12] + h[13]) / waves, ((h[11] + h[12] + h[13]) / waves) / std::max((h[8] + h[9] + h[10]) / lanes, 1e-9));
        }
    } stat_dump{ctx->stream};
#endif
    if (cfg.has_radius && cfg.radius > 0.0f) {   // radius <= 0 finds nothing (nearest_neighbor.rs:255): pure k-NN fallback
        // A radius set that fits the register list is summed in the reference's order (normals_point: radius_fit), so the list
        // is sized for the EXPECTED set -- density of the box x ball volume, + 3 sigma -- when that is within reach of an
        // instantiation, else for k + 1 as before (larger sets then take the f64 moments).
        uint32_t Lw = K1;
        {
            const GridGeom &g = gv.g;
            double vol = 1.0, emax = std::max({(double)g.maxx - g.minx, (double)g.maxy - g.miny, (double)g.maxz - g.minz});
            bool flat = !(emax > 0.0);
            for (double e : {(double)g.maxx - g.minx, (double)g.maxy - g.miny, (double)g.maxz - g.minz}) { if (!(e > 1e-3 * emax)) flat = true; vol *= e; }
            if (!flat && vol > 0.0) {
                const double lam = (double)g.n / vol * 4.18879 * (double)cfg.radius * cfg.radius * cfg.radius;
                const double want = lam + 3.0 * std::sqrt(lam) + 2.0;
                if (want <= 129.0) Lw = std::max<uint32_t>(K1, (uint32_t)want);
            }
        }
        if (Lw <= 11)      launch_variant<11, 256, true>(ctx->stream, gv, prm, d_out6, ctx);
        else if (Lw <= 17) launch_variant<17, 256, true>(ctx->stream, gv, prm, d_out6, ctx);
        else if (Lw <= 33) launch_variant<33, 128, true>(ctx->stream, gv, prm, d_out6, ctx);
        else if (Lw <= 65) launch_variant<65, 64, true>(ctx->stream, gv, prm, d_out6, ctx);
        else               launch_variant<129, 64, true>(ctx->stream, gv, prm, d_out6, ctx);
        TC_HIP_TRY(ctx, hipGetLastError());
        return TC_OK;
    }
    // TC_NORMALS_TAG: the tagged-key path (knn_tagged) -- the list carries the neighbours' positions, no collect pass.  Needs
    // distances that cannot overflow (the keys are compared as integers; an infinite distance with a tag would read as NaN in the
    // pruning radius) and two spare list entries.
    // 0: the register-list kernels; 1 / 2: the tagged path forced, row by row / flattened; 3 (default): flattened, taken where the
    // index is volumetric (NormalParams::tag_policy)
    const int tagged = [] { const char *e = getenv("TC_NORMALS_TAG"); return e ? atoi(e) : 3; }();        // (read per call: the tests flip it)
    prm.tag_policy = tagged >= 3 ? 1 : 0;
    // (a build that read its occupied-cell count back -- edge adaptation, clouds of >= 2^18 points -- lets the HOST apply the same
    // rule and launch the register-list kernels directly: their fallback copy inside the 19-entry tagged kernel runs at five waves
    // per SIMD, 4 % slower on a TUM-shaped 1 M-point surface)
    bool host_says_no = false;
    if (tagged >= 3 && ix.occ_host_valid) {
        const unsigned long long occ = ix.occ_host;
        if (occ * 5ull >= (unsigned long long)ix.geom.ncell && (unsigned long long)ix.geom.n <= occ * 4ull) prm.tag_policy = 0;
        else host_says_no = true;
    }
    if (tagged && !host_says_no && prm.R0 == 2) {
        float ext = 0.0f;
        for (int c = 0; c < 3; ++c) ext = std::max(ext, std::fabs(ix.exact_max[c] - ix.exact_min[c]));
        if (ext < 1e17f && K1 <= 21 && ix.geom.n < (1u << 28)) {
            if (tagged >= 2) {
                if (K1 <= 9)       launch_variant<11, TC_TAG_BLOCK, false, -2>(ctx->stream, gv, prm, d_out6, ctx);
                else if (K1 <= 11) launch_variant<13, TC_TAG_BLOCK, false, -2>(ctx->stream, gv, prm, d_out6, ctx);
                else if (K1 <= 17) launch_variant<19, TC_TAG_BLOCK, false, -2>(ctx->stream, gv, prm, d_out6, ctx);
                else               launch_variant<23, TC_TAG_BLOCK, false, -2>(ctx->stream, gv, prm, d_out6, ctx);
            }
            else if (K1 <= 9)  launch_variant<11, 256, false, -1>(ctx->stream, gv, prm, d_out6, ctx);
            else if (K1 <= 11) launch_variant<13, 256, false, -1>(ctx->stream, gv, prm, d_out6, ctx);
            else if (K1 <= 17) launch_variant<19, 256, false, -1>(ctx->stream, gv, prm, d_out6, ctx);
            else               launch_variant<23, 256, false, -1>(ctx->stream, gv, prm, d_out6, ctx);
            TC_HIP_TRY(ctx, hipGetLastError());
            return TC_OK;
        }
    }
    if (K1 <= 9)       launch_variant<9, 256>(ctx->stream, gv, prm, d_out6, ctx);
    else if (K1 <= 11) launch_variant<11, 256>(ctx->stream, gv, prm, d_out6, ctx);
    else if (K1 <= 17) launch_variant<17, 256>(ctx->stream, gv, prm, d_out6, ctx);
    else if (K1 <= 21) launch_variant<21, 256>(ctx->stream, gv, prm, d_out6, ctx);
    else if (K1 <= 33) launch_variant<33, 128>(ctx->stream, gv, prm, d_out6, ctx);
    else if (K1 <= 65) launch_variant<65, 64>(ctx->stream, gv, prm, d_out6, ctx);
    else               launch_variant<129, 64>(ctx->stream, gv, prm, d_out6, ctx);
    TC_HIP_TRY(ctx, hipGetLastError());
    return TC_OK;
}

}  // namespace tc
