/*
 * tc_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 * See tc_oracle.h for scope and parity status ("bit-level parity unpinned").
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: Rust never contracts a*b+c into an FMA, and the
 * squared-distance expression (nearest_neighbor.rs:162-167) decides neighbour sets.
 *
 * All arithmetic is IEEE f32 in the reference's operation order unless a comment says f64.
 * Citations "xxx.rs:a-b" are relative to /root/reference/threecrate-algorithms/src/.
 */
#include "tc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NIL 0xFFFFFFFFu /* nearest_neighbor.rs:8 */

/* ------------------------------------------------------------------------------------------ */
/* kd-tree: nearest_neighbor.rs:17-159                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float    p[3];
    uint64_t orig;   /* original_index */
    uint32_t left, right;
    uint8_t  axis;
} kd_node;

typedef struct { float p[3]; uint64_t idx; } kd_item; /* (Point3f, usize) work array :46-50 */

struct tco_kdtree {
    kd_node *nodes;
    size_t   n;
    int      has_root;
};

/* Lomuto partition, pivot = last element, "<=" moves left (nearest_neighbor.rs:134-159) */
static size_t kd_partition(kd_item *a, size_t start, size_t end, int axis) {
    float pivot = a[end].p[axis];
    size_t i = start;
    for (size_t j = start; j < end; ++j) {
        if (a[j].p[axis] <= pivot) {
            kd_item t = a[i]; a[i] = a[j]; a[j] = t;
            ++i;
        }
    }
    kd_item t = a[i]; a[i] = a[end]; a[end] = t;
    return i;
}

/* quickselect (nearest_neighbor.rs:112-131) */
static void kd_select_median(kd_item *a, size_t start, size_t end, size_t target, int axis) {
    size_t left = start, right = end;
    while (left < right) {
        size_t p = kd_partition(a, left, right, axis);
        if (p == target) return;
        if (p < target) left = p + 1; else right = p - 1;
    }
}

/* pre-order slot reservation, inclusive bounds (nearest_neighbor.rs:66-109) */
static uint32_t kd_build(kd_node *nodes, size_t *count, kd_item *a, size_t depth, size_t start, size_t end) {
    int axis = (int)(depth % 3);
    size_t median = (start + end) / 2;
    kd_select_median(a, start, end, median, axis);
    uint32_t my = (uint32_t)(*count)++;
    kd_node *nd = &nodes[my];
    nd->p[0] = a[median].p[0]; nd->p[1] = a[median].p[1]; nd->p[2] = a[median].p[2];
    nd->orig = a[median].idx; nd->left = NIL; nd->right = NIL; nd->axis = (uint8_t)axis;
    uint32_t l = NIL, r = NIL;
    if (median > start) l = kd_build(nodes, count, a, depth + 1, start, median - 1);
    if (median < end)   r = kd_build(nodes, count, a, depth + 1, median + 1, end);
    nodes[my].left = l; nodes[my].right = r;
    return my;
}

tco_kdtree *tco_kdtree_new(const float *xyz, size_t n) {
    tco_kdtree *t = (tco_kdtree *)calloc(1, sizeof(*t));
    if (!t) return NULL;
    if (n == 0) return t; /* :38-44 */
    kd_item *a = (kd_item *)malloc(n * sizeof(kd_item));
    t->nodes = (kd_node *)malloc(n * sizeof(kd_node));
    for (size_t i = 0; i < n; ++i) {
        a[i].p[0] = xyz[3 * i]; a[i].p[1] = xyz[3 * i + 1]; a[i].p[2] = xyz[3 * i + 2];
        a[i].idx = i;
    }
    size_t count = 0;
    kd_build(t->nodes, &count, a, 0, 0, n - 1);
    free(a);
    t->n = n; t->has_root = 1;
    return t;
}

void tco_kdtree_free(tco_kdtree *t) { if (t) { free(t->nodes); free(t); } }

/* nearest_neighbor.rs:162-167 -- argument order (node, query) */
static inline float dist_sq(const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}

/* ---- std::collections::BinaryHeap<Neighbor> restated (max-heap on distance; ties Equal) ----
 * Rust std is not under /root/reference; this follows the published library algorithm
 * (sift_up on push; pop = swap-last-to-root, sift_down_to_bottom, sift_up;
 * into_sorted_vec = repeated swap + sift_down_range).  It only influences tie order. */
typedef struct { float d; uint64_t idx; } nb_t;
#define NB_LE(a, b) (!((a).d > (b).d))  /* a <= b under partial_cmp().unwrap_or(Equal) */
#define NB_GE(a, b) (!((a).d < (b).d))
#define NB_LT(a, b) ((a).d < (b).d)

static void heap_sift_up(nb_t *h, size_t start, size_t pos) {
    nb_t e = h[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (NB_LE(e, h[parent])) break;
        h[pos] = h[parent];
        pos = parent;
    }
    h[pos] = e;
}
static void heap_push(nb_t *h, size_t *len, nb_t v) { h[*len] = v; heap_sift_up(h, 0, (*len)++); }
static void heap_sift_down_to_bottom(nb_t *h, size_t end) {
    size_t pos = 0;
    nb_t e = h[0];
    size_t child = 1;
    size_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim && end >= 2) {
        child += NB_LE(h[child], h[child + 1]) ? 1 : 0;
        h[pos] = h[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) { h[pos] = h[child]; pos = child; }
    h[pos] = e;
    heap_sift_up(h, 0, pos);
}
static nb_t heap_pop(nb_t *h, size_t *len) {
    nb_t item = h[--(*len)];
    if (*len > 0) {
        nb_t t = h[0]; h[0] = item; item = t;
        heap_sift_down_to_bottom(h, *len);
    }
    return item;
}
static void heap_sift_down_range(nb_t *h, size_t pos, size_t end) {
    nb_t e = h[pos];
    size_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
        child += NB_LE(h[child], h[child + 1]) ? 1 : 0;
        if (NB_GE(e, h[child])) { h[pos] = e; return; }
        h[pos] = h[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1 && NB_LT(e, h[child])) { h[pos] = h[child]; pos = child; }
    h[pos] = e;
}
static void heap_into_sorted(nb_t *h, size_t len) {
    size_t end = len;
    while (end > 1) {
        --end;
        nb_t t = h[0]; h[0] = h[end]; h[end] = t;
        heap_sift_down_range(h, 0, end);
    }
}

/* scratch for one query thread */
typedef struct { nb_t *heap; uint32_t *stack; size_t heap_cap, stack_cap; } kd_scratch;
static void scratch_init(kd_scratch *s) { memset(s, 0, sizeof(*s)); }
static void scratch_free(kd_scratch *s) { free(s->heap); free(s->stack); }
static void scratch_reserve(kd_scratch *s, size_t k) {
    if (s->heap_cap < k + 2) { s->heap_cap = k + 2; s->heap = (nb_t *)realloc(s->heap, s->heap_cap * sizeof(nb_t)); }
    if (s->stack_cap < 256) { s->stack_cap = 256; s->stack = (uint32_t *)realloc(s->stack, s->stack_cap * sizeof(uint32_t)); }
}
static inline void stack_push(kd_scratch *s, size_t *sp, uint32_t v) {
    if (*sp == s->stack_cap) { s->stack_cap *= 2; s->stack = (uint32_t *)realloc(s->stack, s->stack_cap * sizeof(uint32_t)); }
    s->stack[(*sp)++] = v;
}

/* find_k_nearest (nearest_neighbor.rs:177-251). Leaves sorted (d2, idx) in s->heap; returns count. */
static size_t kd_knn_core(const tco_kdtree *t, const float q[3], size_t k, kd_scratch *s) {
    if (k == 0 || t->n == 0) return 0;
    scratch_reserve(s, k);
    nb_t *heap = s->heap;
    size_t hlen = 0, sp = 0;
    if (t->has_root) stack_push(s, &sp, 0);
    while (sp > 0) {
        uint32_t idx = s->stack[--sp];
        const kd_node *node = &t->nodes[idx];
        float d2 = dist_sq(node->p, q);
        if (hlen < k) {
            nb_t v = { d2, node->orig }; heap_push(heap, &hlen, v);
        } else if (d2 < heap[0].d) {
            heap_pop(heap, &hlen);
            nb_t v = { d2, node->orig }; heap_push(heap, &hlen, v);
        }
        float qv = q[node->axis], nv = node->p[node->axis];
        float axis_dist = qv - nv;
        float axis_dist_sq = axis_dist * axis_dist;
        uint32_t near_c, far_c;
        if (qv <= nv) { near_c = node->left; far_c = node->right; }
        else          { near_c = node->right; far_c = node->left; }
        int search_far = (hlen > 0) ? (hlen < k || axis_dist_sq < heap[0].d) : 1;
        if (search_far && far_c != NIL) stack_push(s, &sp, far_c);
        if (near_c != NIL) stack_push(s, &sp, near_c);
    }
    heap_into_sorted(heap, hlen);
    return hlen;
}

size_t tco_kdtree_knn(const tco_kdtree *t, const float q[3], size_t k, uint64_t *idx, float *dist) {
    kd_scratch s; scratch_init(&s);
    size_t m = kd_knn_core(t, q, k, &s);
    for (size_t i = 0; i < m; ++i) { idx[i] = s.heap[i].idx; dist[i] = sqrtf(s.heap[i].d); }
    scratch_free(&s);
    return m;
}

static int cmp_nb_dist(const void *a, const void *b) {
    float x = ((const nb_t *)a)->d, y = ((const nb_t *)b)->d;
    return (x < y) ? -1 : (x > y) ? 1 : 0;
}
/* stable sort by distance (Rust sort_by is a stable merge sort) */
static void stable_sort_nb(nb_t *a, size_t n) {
    if (n < 2) return;
    nb_t *tmp = (nb_t *)malloc(n * sizeof(nb_t));
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) tmp[o++] = (cmp_nb_dist(&a[j], &a[i]) < 0) ? a[j++] : a[i++];
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(nb_t));
    }
    free(tmp);
}

/* find_radius_neighbors (nearest_neighbor.rs:254-298); result (idx, sqrt(d2)) sorted by distance.
   Returns malloc'd array in *out (caller frees). */
static size_t kd_radius_core(const tco_kdtree *t, const float q[3], float radius, nb_t **out, kd_scratch *s) {
    *out = NULL;
    if (radius <= 0.0f || t->n == 0) return 0;
    float r2 = radius * radius;
    scratch_reserve(s, 1);
    size_t cap = 32, cnt = 0, sp = 0;
    nb_t *res = (nb_t *)malloc(cap * sizeof(nb_t));
    if (t->has_root) stack_push(s, &sp, 0);
    while (sp > 0) {
        uint32_t idx = s->stack[--sp];
        const kd_node *node = &t->nodes[idx];
        float d2 = dist_sq(node->p, q);
        if (d2 <= r2) {
            if (cnt == cap) { cap *= 2; res = (nb_t *)realloc(res, cap * sizeof(nb_t)); }
            res[cnt].idx = node->orig; res[cnt].d = sqrtf(d2); ++cnt;
        }
        float qv = q[node->axis], nv = node->p[node->axis];
        float axis_dist = qv - nv;
        uint32_t near_c, far_c;
        if (qv <= nv) { near_c = node->left; far_c = node->right; }
        else          { near_c = node->right; far_c = node->left; }
        if (axis_dist * axis_dist <= r2) { if (far_c != NIL) stack_push(s, &sp, far_c); }
        if (near_c != NIL) stack_push(s, &sp, near_c);
    }
    stable_sort_nb(res, cnt);
    *out = res;
    return cnt;
}

size_t tco_kdtree_radius(const tco_kdtree *t, const float q[3], float radius,
                         uint64_t *idx, float *dist, size_t cap) {
    kd_scratch s; scratch_init(&s);
    nb_t *res; size_t m = kd_radius_core(t, q, radius, &res, &s);
    for (size_t i = 0; i < m && i < cap; ++i) { idx[i] = res[i].idx; dist[i] = res[i].d; }
    free(res); scratch_free(&s);
    return m;
}

/* BruteForceSearch::find_k_nearest (nearest_neighbor.rs:340-362) */
size_t tco_brute_knn(const float *xyz, size_t n, const float q[3], size_t k, uint64_t *idx, float *dist) {
    if (k == 0 || n == 0) return 0;
    nb_t *a = (nb_t *)malloc(n * sizeof(nb_t));
    for (size_t i = 0; i < n; ++i) {
        float dx = xyz[3 * i] - q[0], dy = xyz[3 * i + 1] - q[1], dz = xyz[3 * i + 2] - q[2];
        a[i].idx = i; a[i].d = sqrtf(dx * dx + dy * dy + dz * dz);
    }
    stable_sort_nb(a, n);
    size_t m = k < n ? k : n;
    for (size_t i = 0; i < m; ++i) { idx[i] = a[i].idx; dist[i] = a[i].d; }
    free(a);
    return m;
}

/* threads a call with `threads <= 0` uses: OpenMP's maximum, capped by tco_set_max_threads (oracle.py passes the container's
   CPU quota: a box that shows 256 CPUs under a 16-core quota runs 256 runnable threads 10x slower than 16) */
static int g_max_threads = 0;
void tco_set_max_threads(int n) { g_max_threads = n > 0 ? n : 0; }
int tco_num_threads(void) {
#ifdef _OPENMP
    int mx = omp_get_max_threads();
    return (g_max_threads > 0 && g_max_threads < mx) ? g_max_threads : mx;
#else
    return 1;
#endif
}
static int resolve_threads(int threads) {
    int mx = tco_num_threads();
    if (threads <= 0 || threads > mx) return mx;
    return threads;
}

int tco_knn_batch(const float *xyz, size_t n, const float *queries, size_t nq, size_t k,
                  uint64_t *idx, float *dist, uint32_t *counts, int threads) {
    tco_kdtree *t = tco_kdtree_new(xyz, n);
    int nt = resolve_threads(threads);
    (void)nt;
#pragma omp parallel num_threads(nt)
    {
        kd_scratch s; scratch_init(&s);
#pragma omp for schedule(dynamic, 256)
        for (long long i = 0; i < (long long)nq; ++i) {
            size_t m = kd_knn_core(t, &queries[3 * i], k, &s);
            for (size_t j = 0; j < m; ++j) { idx[i * k + j] = s.heap[j].idx; dist[i * k + j] = sqrtf(s.heap[j].d); }
            counts[i] = (uint32_t)m;
        }
        scratch_free(&s);
    }
    tco_kdtree_free(t);
    return TCO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* nalgebra 0.34 small dense algebra, restated from its published algorithms (f32)              */
/* ------------------------------------------------------------------------------------------ */
#define EPS32 1.1920929e-07f

/* ComplexField::to_exp for reals: (|x|, x/|x|), (0, 1) for zero */
static inline void to_exp(float x, float *mod, float *sign) {
    float n = fabsf(x);
    if (n != 0.0f) { *mod = n; *sign = x / n; } else { *mod = 0.0f; *sign = 1.0f; }
}

/* linalg::givens::GivensRotation::cancel_y */
static int givens_cancel_y(float x, float y, float *c, float *s, float *r) {
    if (y != 0.0f) {
        float mod0, sign0; to_exp(x, &mod0, &sign0);
        float denom = sqrtf(mod0 * mod0 + y * y);
        *c = mod0 / denom;
        *s = -y / (sign0 * denom);
        *r = sign0 * denom;
        return 1;
    }
    return 0;
}

/* linalg::symmetric_eigen::wilkinson_shift */
static float wilkinson_shift(float tmm, float tnn, float tmn) {
    float sq = tmn * tmn;
    if (sq != 0.0f) {
        float d = (tmm - tnn) * 0.5f;
        float sg = (d >= 0.0f || d != d) ? 1.0f : -1.0f; /* f32::signum(+0)=1 */
        if (d == 0.0f && signbit(d)) sg = -1.0f;
        return tnn - sq / (d + sg * sqrtf(d * d + sq));
    }
    return tnn;
}

static void se_delimit(const float diag[3], float off[2], size_t end, float eps, size_t *start_out, size_t *end_out) {
    size_t n = end;
    while (n > 0) {
        size_t m = n - 1;
        if (fabsf(off[m]) > eps * (fabsf(diag[n]) + fabsf(diag[m]))) break;
        --n;
    }
    if (n == 0) { *start_out = 0; *end_out = 0; return; }
    size_t ns = n - 1;
    while (ns > 0) {
        size_t m = ns - 1;
        if (off[m] == 0.0f || fabsf(off[m]) <= eps * (fabsf(diag[ns]) + fabsf(diag[m]))) { off[m] = 0.0f; break; }
        --ns;
    }
    *start_out = ns; *end_out = n;
}

/*
 * Matrix3::symmetric_eigen (used at normals.rs:181):
 *   scale by max-abs entry, Householder tridiagonalisation (SymmetricTridiagonal),
 *   implicit symmetric QR with Wilkinson shifts, direct 2x2 solve for the last block;
 *   eigenvalues UNSORTED, eigenvectors = columns of q.
 * a is symmetric row-major; only the lower triangle is read (like nalgebra's hegemv/hegerc).
 */
static void sym_eigen3(const float a_in[3][3], float evals[3], float q[3][3]) {
    float a[3][3];
    float amax = 0.0f;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = a_in[i][j]; float v = fabsf(a[i][j]); if (v > amax) amax = v; }
    if (amax != 0.0f) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = a[i][j] / amax;

    float diag[3], off[2], offs[2];
    float u[2] = { 0.0f, 0.0f };
    int refl0 = 0;
    {   /* step i = 0: reflect (a10, a20) onto e1 */
        float x0 = a[1][0], x1 = a[2][0];
        float sq = x0 * x0 + x1 * x1;
        float nrm = sqrtf(sq);
        float mod, sign; to_exp(x0, &mod, &sign);
        float signed_norm = sign * nrm;
        float factor = (sq + mod * nrm) * 2.0f;
        if (factor != 0.0f) {
            float f = sqrtf(factor);
            u[0] = (x0 + signed_norm) / f; u[1] = x1 / f;
            float un = sqrtf(u[0] * u[0] + u[1] * u[1]);
            u[0] /= un; u[1] /= un;
            offs[0] = -signed_norm; refl0 = 1;
            /* B <- H B H on the trailing 2x2 (lower triangle b00=a11, b10=a21, b11=a22) */
            float b00 = a[1][1], b10 = a[2][1], b11 = a[2][2];
            float p0 = 2.0f * (b00 * u[0] + b10 * u[1]);
            float p1 = 2.0f * (b10 * u[0] + b11 * u[1]);
            float dot = u[0] * p0 + u[1] * p1;
            b00 = b00 - p0 * u[0]; b10 = b10 - p1 * u[0]; b11 = b11 - p1 * u[1];
            b00 = b00 - u[0] * p0; b10 = b10 - u[1] * p0; b11 = b11 - u[1] * p1;
            float d2 = dot * 2.0f;
            b00 = b00 + d2 * u[0] * u[0]; b10 = b10 + d2 * u[1] * u[0]; b11 = b11 + d2 * u[1] * u[1];
            a[1][1] = b00; a[2][1] = b10; a[2][2] = b11;
        } else {
            offs[0] = signed_norm;
        }
    }
    float ax1 = 0.0f; int refl1 = 0;
    {   /* step i = 1: 1-vector (a21) */
        float x0 = a[2][1];
        float sq = x0 * x0, nrm = sqrtf(sq);
        float mod, sign; to_exp(x0, &mod, &sign);
        float signed_norm = sign * nrm;
        float factor = (sq + mod * nrm) * 2.0f;
        if (factor != 0.0f) {
            float f = sqrtf(factor);
            ax1 = (x0 + signed_norm) / f;
            ax1 = ax1 / fabsf(ax1);
            offs[1] = -signed_norm; refl1 = 1;
            /* 1x1 trailing block is invariant under the reflection (up to rounding): p=2*b*ax; */
            float b = a[2][2];
            float p = 2.0f * (b * ax1);
            float dot = ax1 * p;
            b = b - p * ax1; b = b - ax1 * p; b = b + (dot * 2.0f) * ax1 * ax1;
            a[2][2] = b;
        } else {
            offs[1] = signed_norm;
        }
    }
    diag[0] = a[0][0]; diag[1] = a[1][1]; diag[2] = a[2][2];
    off[0] = fabsf(offs[0]); off[1] = fabsf(offs[1]);

    /* householder::assemble_q with signs = off_diagonal (before modulus) */
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) q[i][j] = (i == j) ? 1.0f : 0.0f;
    if (refl1 || 1) {
        float sg = (offs[1] < 0.0f || (offs[1] == 0.0f && signbit(offs[1]))) ? -1.0f : 1.0f;
        float axis = refl1 ? ax1 : 0.0f;
        for (int c = 1; c < 3; ++c) {           /* rows 2.., cols 1.. */
            float col = q[2][c];
            float factor = (axis * col) * -2.0f;
            q[2][c] = sg * col + axis * (factor * sg);
        }
        if (!refl1) { /* zero axis: reflect_with_sign still scales by sign */ }
    }
    {
        float sg = (offs[0] < 0.0f || (offs[0] == 0.0f && signbit(offs[0]))) ? -1.0f : 1.0f;
        float ax[2] = { refl0 ? u[0] : 0.0f, refl0 ? u[1] : 0.0f };
        for (int c = 0; c < 3; ++c) {           /* rows 1.., cols 0.. */
            float c0 = q[1][c], c1 = q[2][c];
            float factor = (ax[0] * c0 + ax[1] * c1) * -2.0f;
            q[1][c] = sg * c0 + ax[0] * (factor * sg);
            q[2][c] = sg * c1 + ax[1] * (factor * sg);
        }
    }

    /* implicit QR iterations (SymmetricEigen::do_decompose) */
    const float eps = EPS32;
    size_t start, end;
    se_delimit(diag, off, 2, eps, &start, &end);
    int guard = 0;
    while (end != start && guard++ < 10000) {
        size_t subdim = end - start + 1;
        if (subdim > 2) {
            size_t m = end - 1, n = end;
            float vx = diag[start] - wilkinson_shift(diag[m], diag[n], off[m]);
            float vy = off[start];
            for (size_t i = start; i < n; ++i) {
                size_t j = i + 1;
                float c, s, nrm;
                if (!givens_cancel_y(vx, vy, &c, &s, &nrm)) break;
                if (i > start) off[i - 1] = nrm;
                float mii = diag[i], mjj = diag[j], mij = off[i];
                float cc = c * c, ss = s * s, cs = c * s;
                float b = cs * 2.0f * mij;
                diag[i] = (cc * mii + ss * mjj) - b;
                diag[j] = (ss * mii + cc * mjj) + b;
                off[i] = cs * (mii - mjj) + mij * (cc - ss);
                if (i != n - 1) {
                    vx = off[i];
                    vy = -s * off[i + 1];
                    off[i + 1] *= c;
                }
                /* q <- q * G, G = [[c, s], [-s, c]] on columns (i, j) */
                for (int r = 0; r < 3; ++r) {
                    float qa = q[r][i], qb = q[r][j];
                    q[r][i] = qa * c - s * qb;
                    q[r][j] = s * qa + qb * c;
                }
            }
            if (fabsf(off[m]) <= eps * (fabsf(diag[m]) + fabsf(diag[n]))) end -= 1;
        } else if (subdim == 2) {
            float h00 = diag[start], h10 = off[start], h11 = diag[start + 1];
            float val = (h00 - h11) * 0.5f;
            float discr = h10 * h10 + val * val;
            float sq = sqrtf(discr);
            float half_tra = (h00 + h11) * 0.5f;
            float e0 = half_tra + sq, e1 = half_tra - sq;
            float bx = e0 - diag[start + 1], by = off[start];
            diag[start] = e0; diag[start + 1] = e1;
            /* GivensRotation::try_new(bx, by, eps) */
            float mod0, sign0; to_exp(bx, &mod0, &sign0);
            float denom = sqrtf(mod0 * mod0 + by * by);
            if (denom > eps) {
                float c = mod0 / denom, s = by / (sign0 * denom);
                for (int r = 0; r < 3; ++r) {
                    float qa = q[r][start], qb = q[r][start + 1];
                    q[r][start] = qa * c + s * qb;
                    q[r][start + 1] = -s * qa + qb * c;
                }
            }
            end -= 1;
        }
        se_delimit(diag, off, end, eps, &start, &end);
    }
    for (int i = 0; i < 3; ++i) evals[i] = diag[i] * amax;
}

void tco_symmetric_eigen3(const float m[9], float evals[3], float evecs[9]) {
    float a[3][3], q[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = m[3 * i + j];
    sym_eigen3(a, evals, q);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) evecs[3 * i + j] = q[i][j];
}

/*
 * Matrix3::svd(true, true) (registration.rs:175): Golub-Kahan bidiagonalisation + implicit
 * shifted QR on the bidiagonal, singular values made non-negative and sorted descending with
 * U / V^T permuted consistently.  (nalgebra's exact rotation bookkeeping is not reproduced;
 * R = V U^T is invariant to it for non-degenerate H.)  f32 throughout.
 */
static float pythag32(float a, float b) {
    float aa = fabsf(a), ab = fabsf(b);
    if (aa > ab) { float r = ab / aa; return aa * sqrtf(1.0f + r * r); }
    if (ab == 0.0f) return 0.0f;
    float r = aa / ab; return ab * sqrtf(1.0f + r * r);
}
static float sign32(float a, float b) { return (b >= 0.0f) ? fabsf(a) : -fabsf(a); }

static void svd3(const float m_in[3][3], float U[3][3], float w[3], float Vt[3][3]) {
    enum { N = 3 };
    float a[N][N], v[N][N], rv1[N];
    float amax = 0.0f;
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { a[i][j] = m_in[i][j]; float t = fabsf(a[i][j]); if (t > amax) amax = t; }
    if (amax != 0.0f) for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) a[i][j] /= amax;
    float g = 0.0f, scale = 0.0f, anorm = 0.0f;
    int l = 0;
    /* Householder reduction to bidiagonal form */
    for (int i = 0; i < N; ++i) {
        l = i + 1;
        rv1[i] = scale * g;
        g = 0.0f; scale = 0.0f;
        float s = 0.0f;
        for (int k = i; k < N; ++k) scale += fabsf(a[k][i]);
        if (scale != 0.0f) {
            for (int k = i; k < N; ++k) { a[k][i] /= scale; s += a[k][i] * a[k][i]; }
            float f = a[i][i];
            g = -sign32(sqrtf(s), f);
            float h = f * g - s;
            a[i][i] = f - g;
            for (int j = l; j < N; ++j) {
                float ss = 0.0f;
                for (int k = i; k < N; ++k) ss += a[k][i] * a[k][j];
                float ff = ss / h;
                for (int k = i; k < N; ++k) a[k][j] += ff * a[k][i];
            }
            for (int k = i; k < N; ++k) a[k][i] *= scale;
        }
        w[i] = scale * g;
        g = 0.0f; s = 0.0f; scale = 0.0f;
        if (i < N && i != N - 1) {
            for (int k = l; k < N; ++k) scale += fabsf(a[i][k]);
            if (scale != 0.0f) {
                for (int k = l; k < N; ++k) { a[i][k] /= scale; s += a[i][k] * a[i][k]; }
                float f = a[i][l];
                g = -sign32(sqrtf(s), f);
                float h = f * g - s;
                a[i][l] = f - g;
                for (int k = l; k < N; ++k) rv1[k] = a[i][k] / h;
                for (int j = l; j < N; ++j) {
                    float ss = 0.0f;
                    for (int k = l; k < N; ++k) ss += a[j][k] * a[i][k];
                    for (int k = l; k < N; ++k) a[j][k] += ss * rv1[k];
                }
                for (int k = l; k < N; ++k) a[i][k] *= scale;
            }
        }
        float t = fabsf(w[i]) + fabsf(rv1[i]);
        if (t > anorm) anorm = t;
    }
    /* accumulate right-hand transformations */
    for (int i = N - 1; i >= 0; --i) {
        if (i < N - 1) {
            if (g != 0.0f) {
                for (int j = l; j < N; ++j) v[j][i] = (a[i][j] / a[i][l]) / g;
                for (int j = l; j < N; ++j) {
                    float s = 0.0f;
                    for (int k = l; k < N; ++k) s += a[i][k] * v[k][j];
                    for (int k = l; k < N; ++k) v[k][j] += s * v[k][i];
                }
            }
            for (int j = l; j < N; ++j) { v[i][j] = 0.0f; v[j][i] = 0.0f; }
        }
        v[i][i] = 1.0f;
        g = rv1[i];
        l = i;
    }
    /* accumulate left-hand transformations */
    for (int i = N - 1; i >= 0; --i) {
        l = i + 1;
        g = w[i];
        for (int j = l; j < N; ++j) a[i][j] = 0.0f;
        if (g != 0.0f) {
            g = 1.0f / g;
            for (int j = l; j < N; ++j) {
                float s = 0.0f;
                for (int k = l; k < N; ++k) s += a[k][i] * a[k][j];
                float f = (s / a[i][i]) * g;
                for (int k = i; k < N; ++k) a[k][j] += f * a[k][i];
            }
            for (int j = i; j < N; ++j) a[j][i] *= g;
        } else {
            for (int j = i; j < N; ++j) a[j][i] = 0.0f;
        }
        a[i][i] += 1.0f;
    }
    /* diagonalisation of the bidiagonal form */
    for (int k = N - 1; k >= 0; --k) {
        for (int its = 0; its < 60; ++its) {
            int flag = 1, nm = 0;
            for (l = k; l >= 0; --l) {
                nm = l - 1;
                if (fabsf(rv1[l]) <= EPS32 * anorm) { flag = 0; break; }
                if (fabsf(w[nm]) <= EPS32 * anorm) break;
            }
            if (flag) {
                float c = 0.0f, s = 1.0f;
                for (int i = l; i <= k; ++i) {
                    float f = s * rv1[i];
                    rv1[i] = c * rv1[i];
                    if (fabsf(f) <= EPS32 * anorm) break;
                    g = w[i];
                    float h = pythag32(f, g);
                    w[i] = h;
                    h = 1.0f / h;
                    c = g * h; s = -f * h;
                    for (int j = 0; j < N; ++j) {
                        float y = a[j][nm], z = a[j][i];
                        a[j][nm] = y * c + z * s;
                        a[j][i] = z * c - y * s;
                    }
                }
            }
            float z = w[k];
            if (l == k) {
                if (z < 0.0f) { w[k] = -z; for (int j = 0; j < N; ++j) v[j][k] = -v[j][k]; }
                break;
            }
            float x = w[l];
            nm = k - 1;
            float y = w[nm];
            g = rv1[nm];
            float h = rv1[k];
            float f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0f * h * y);
            g = pythag32(f, 1.0f);
            f = ((x - z) * (x + z) + h * ((y / (f + sign32(g, f))) - h)) / x;
            float c = 1.0f, s = 1.0f;
            for (int j = l; j <= nm; ++j) {
                int i = j + 1;
                g = rv1[i];
                y = w[i];
                h = s * g;
                g = c * g;
                z = pythag32(f, h);
                rv1[j] = z;
                c = f / z; s = h / z;
                f = x * c + g * s;
                g = g * c - x * s;
                h = y * s;
                y *= c;
                for (int jj = 0; jj < N; ++jj) {
                    float xx = v[jj][j], zz = v[jj][i];
                    v[jj][j] = xx * c + zz * s;
                    v[jj][i] = zz * c - xx * s;
                }
                z = pythag32(f, h);
                w[j] = z;
                if (z != 0.0f) { z = 1.0f / z; c = f * z; s = h * z; }
                f = c * g + s * y;
                x = c * y - s * g;
                for (int jj = 0; jj < N; ++jj) {
                    float yy = a[jj][j], zz = a[jj][i];
                    a[jj][j] = yy * c + zz * s;
                    a[jj][i] = zz * c - yy * s;
                }
            }
            rv1[l] = 0.0f;
            rv1[k] = f;
            w[k] = x;
        }
    }
    /* sort descending (SVD::sort_by_singular_values), permuting U columns / V^T rows */
    int order[3] = { 0, 1, 2 };
    for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j)
        if (w[order[j]] > w[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    float ws[3];
    for (int c = 0; c < 3; ++c) {
        ws[c] = w[order[c]] * amax;
        for (int r = 0; r < 3; ++r) { U[r][c] = a[r][order[c]]; Vt[c][r] = v[r][order[c]]; }
    }
    w[0] = ws[0]; w[1] = ws[1]; w[2] = ws[2];
}

void tco_svd3(const float m[9], float u[9], float s[3], float vt[9]) {
    float a[3][3], U[3][3], Vt[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = m[3 * i + j];
    svd3(a, U, s, Vt);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { u[3 * i + j] = U[i][j]; vt[3 * i + j] = Vt[i][j]; }
}

/* 3x3 product c = a*b, k-sequential accumulation (column axpy order of nalgebra's small gemm) */
static void mat3_mul(const float a[3][3], const float b[3][3], float c[3][3]) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        c[i][j] = (a[i][0] * b[0][j] + a[i][1] * b[1][j]) + a[i][2] * b[2][j];
}
static float mat3_det(const float m[3][3]) {
    float minor_12_23 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float minor_11_23 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float minor_11_22 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    return m[0][0] * minor_12_23 - m[0][1] * minor_11_23 + m[0][2] * minor_11_22;
}

static inline void cross3(const float a[3], const float b[3], float o[3]) {
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Rotation3::from_axis_angle (unit axis) */
static void rot_from_axis_angle(const float u[3], float angle, float r[3][3]) {
    float ux = u[0], uy = u[1], uz = u[2];
    float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    float sn = sinf(angle), cs = cosf(angle);
    float omc = 1.0f - cs;
    r[0][0] = sqx + (1.0f - sqx) * cs;   r[0][1] = ux * uy * omc - uz * sn; r[0][2] = ux * uz * omc + uy * sn;
    r[1][0] = ux * uy * omc + uz * sn;   r[1][1] = sqy + (1.0f - sqy) * cs; r[1][2] = uy * uz * omc - ux * sn;
    r[2][0] = ux * uz * omc - uy * sn;   r[2][1] = uy * uz * omc + ux * sn; r[2][2] = sqz + (1.0f - sqz) * cs;
}

/* UnitQuaternion::from_rotation_matrix; q = (i, j, k, w) */
static void quat_from_rotmat(const float m[3][3], float q[4]) {
    float tr = m[0][0] + m[1][1] + m[2][2];
    float w, i, j, k;
    if (tr > 0.0f) {
        float denom = sqrtf(tr + 1.0f) * 2.0f;
        w = 0.25f * denom; i = (m[2][1] - m[1][2]) / denom; j = (m[0][2] - m[2][0]) / denom; k = (m[1][0] - m[0][1]) / denom;
    } else if (m[0][0] > m[1][1] && m[0][0] > m[2][2]) {
        float denom = sqrtf(1.0f + m[0][0] - m[1][1] - m[2][2]) * 2.0f;
        w = (m[2][1] - m[1][2]) / denom; i = 0.25f * denom; j = (m[0][1] + m[1][0]) / denom; k = (m[0][2] + m[2][0]) / denom;
    } else if (m[1][1] > m[2][2]) {
        float denom = sqrtf(1.0f + m[1][1] - m[0][0] - m[2][2]) * 2.0f;
        w = (m[0][2] - m[2][0]) / denom; i = (m[0][1] + m[1][0]) / denom; j = 0.25f * denom; k = (m[1][2] + m[2][1]) / denom;
    } else {
        float denom = sqrtf(1.0f + m[2][2] - m[0][0] - m[1][1]) * 2.0f;
        w = (m[1][0] - m[0][1]) / denom; i = (m[0][2] + m[2][0]) / denom; j = (m[1][2] + m[2][1]) / denom; k = 0.25f * denom;
    }
    q[0] = i; q[1] = j; q[2] = k; q[3] = w;
}

static float mat3_diff_norm_sq(const float a[3][3], const float b[3][3]) {
    float s = 0.0f;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { float d = a[r][c] - b[r][c]; s += d * d; }
    return s;
}

/*
 * UnitQuaternion::from_matrix (registration.rs:194) = Rotation3::from_matrix_eps(m, EPSILON, 0, I):
 * iterative closest-rotation extraction (Mueller et al.), then from_rotation_matrix.
 */
static void quat_from_matrix(const float m[3][3], float q[4]) {
    const float eps = EPS32;
    float eps_dist = sqrtf(eps); if (eps * eps > eps_dist) eps_dist = eps * eps;
    float paxes[3] = { 1.0f, 0.0f, 0.0f };
    float rot[3][3] = { {1, 0, 0}, {0, 1, 0}, {0, 0, 1} };
    for (long it = 0; it < 100000; ++it) {
        float rc[3][3], mc[3][3]; /* columns */
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { rc[c][r] = rot[r][c]; mc[c][r] = m[r][c]; }
        float c0[3], c1[3], c2[3];
        cross3(rc[0], mc[0], c0); cross3(rc[1], mc[1], c1); cross3(rc[2], mc[2], c2);
        float axis[3] = { c0[0] + c1[0] + c2[0], c0[1] + c1[1] + c2[1], c0[2] + c1[2] + c2[2] };
        float denom = dot3(rc[0], mc[0]) + dot3(rc[1], mc[1]) + dot3(rc[2], mc[2]);
        float dd = fabsf(denom) + eps;
        float aa[3] = { axis[0] / dd, axis[1] / dd, axis[2] / dd };
        float sqn = dot3(aa, aa);
        if (sqn > eps * eps) {
            float n = sqrtf(sqn);
            float ua[3] = { aa[0] / n, aa[1] / n, aa[2] / n };
            float d[3][3], nr[3][3];
            rot_from_axis_angle(ua, n, d);
            mat3_mul(d, rot, nr);
            memcpy(rot, nr, sizeof(nr));
        } else {
            float pert[3][3]; memcpy(pert, rot, sizeof(pert));
            float nsq = mat3_diff_norm_sq(m, rot), nnsq = nsq;
            for (int g = 0; g < 1000; ++g) {
                float d[3][3], np[3][3];
                rot_from_axis_angle(paxes, eps_dist, d);
                mat3_mul(pert, d, np);
                memcpy(pert, np, sizeof(np));
                nnsq = mat3_diff_norm_sq(m, pert);
                if (fabsf(nsq - nnsq) > eps) break;
            }
            if (nsq < nnsq) break;
            float t = paxes[0]; paxes[0] = paxes[1]; paxes[1] = paxes[2]; paxes[2] = t; /* yzx */
            memcpy(rot, pert, sizeof(pert));
        }
    }
    quat_from_rotmat(rot, q);
}

void tco_quat_from_matrix(const float r[9], float q[4]) {
    float m[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = r[3 * i + j];
    quat_from_matrix(m, q);
}

/* UnitQuaternion * Vector3: t = 2 (qv x p); p' = t*w + qv x t + p */
static inline void quat_rotate(const float q[4], const float p[3], float o[3]) {
    float t[3], c[3];
    cross3(q, p, t);
    t[0] = t[0] * 2.0f; t[1] = t[1] * 2.0f; t[2] = t[2] * 2.0f;
    cross3(q, t, c);
    o[0] = (t[0] * q[3] + c[0]) + p[0];
    o[1] = (t[1] * q[3] + c[1]) + p[1];
    o[2] = (t[2] * q[3] + c[2]) + p[2];
}
/* Isometry3 * Point3 : rotate then translate; T = (qi qj qk qw tx ty tz) */
void tco_isometry_apply(const float T[7], const float p[3], float out[3]) {
    float r[3]; quat_rotate(T, p, r);
    out[0] = r[0] + T[4]; out[1] = r[1] + T[5]; out[2] = r[2] + T[6];
}
/* Quaternion Hamilton product, storage (i j k w) */
static void quat_mul(const float a[4], const float b[4], float o[4]) {
    float w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    float i = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    float j = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    float k = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[0] = i; o[1] = j; o[2] = k; o[3] = w;
}
/* Isometry3 * Isometry3: t = t_a + R_a t_b ; q = q_a q_b (no renormalisation) */
void tco_isometry_mul(const float a[7], const float b[7], float out[7]) {
    float sh[3]; quat_rotate(a, &b[4], sh);
    float q[4]; quat_mul(a, b, q);
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
    out[4] = a[4] + sh[0]; out[5] = a[5] + sh[1]; out[6] = a[6] + sh[2];
}
/* Isometry3::to_homogeneous (threecrate-python/src/lib.rs:48-61 uses it); row-major 4x4 */
void tco_isometry_to_matrix(const float T[7], float m[16]) {
    float i = T[0], j = T[1], k = T[2], w = T[3];
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    m[0] = ww + ii - jj - kk; m[1] = ij - wk;           m[2] = wj + ik;            m[3] = T[4];
    m[4] = wk + ij;           m[5] = ww - ii + jj - kk; m[6] = jk - wi;            m[7] = T[5];
    m[8] = ik - wj;           m[9] = wi + jk;           m[10] = ww - ii - jj + kk; m[11] = T[6];
    m[12] = 0; m[13] = 0; m[14] = 0; m[15] = 1;
}

/* Cholesky::new + solve (6x6, left-looking column form, None on a non-positive pivot) */
int tco_cholesky6_solve(const float a_in[36], const float b[6], float x[6]) {
    float a[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) a[i][j] = a_in[6 * i + j];
    for (int j = 0; j < 6; ++j) {
        for (int k = 0; k < j; ++k) {
            float factor = -a[j][k];
            for (int r = j; r < 6; ++r) a[r][j] = factor * a[r][k] + a[r][j];
        }
        float d = a[j][j];
        if (d != 0.0f && d >= 0.0f) {
            float denom = sqrtf(d);
            a[j][j] = denom;
            for (int r = j + 1; r < 6; ++r) a[r][j] /= denom;
            continue;
        }
        return 0;
    }
    for (int i = 0; i < 6; ++i) x[i] = b[i];
    /* L y = b (column-oriented forward substitution) */
    for (int i = 0; i < 6; ++i) {
        x[i] = x[i] / a[i][i];
        float c = -x[i];
        for (int r = i + 1; r < 6; ++r) x[r] = c * a[r][i] + x[r];
    }
    /* L^T x = y */
    for (int i = 5; i >= 0; --i) {
        float dot = 0.0f;
        for (int r = i + 1; r < 6; ++r) dot += a[r][i] * x[r];
        x[i] = (x[i] - dot) / a[i][i];
    }
    return 1;
}

/* LU::new (partial pivoting) + solve; 0 when a diagonal of U is zero */
int tco_lu6_solve(const float a_in[36], const float b[6], float x[6]) {
    float a[6][6]; int perm[6];
    for (int i = 0; i < 6; ++i) { perm[i] = i; for (int j = 0; j < 6; ++j) a[i][j] = a_in[6 * i + j]; }
    for (int i = 0; i < 6; ++i) x[i] = b[i];
    for (int i = 0; i < 6; ++i) {
        int piv = i; float best = fabsf(a[i][i]);
        for (int r = i + 1; r < 6; ++r) { float v = fabsf(a[r][i]); if (v > best) { best = v; piv = r; } }
        float diag = a[piv][i];
        if (diag == 0.0f) continue;
        if (piv != i) {
            for (int c = 0; c < 6; ++c) { float t = a[i][c]; a[i][c] = a[piv][c]; a[piv][c] = t; }
            float t = x[i]; x[i] = x[piv]; x[piv] = t;
        }
        float inv = 1.0f / diag;
        for (int r = i + 1; r < 6; ++r) {
            a[r][i] *= inv;
            float f = a[r][i];
            for (int c = i + 1; c < 6; ++c) a[r][c] -= f * a[i][c];
        }
    }
    (void)perm;
    for (int i = 0; i < 6; ++i) for (int r = i + 1; r < 6; ++r) x[r] -= a[r][i] * x[i];
    for (int i = 5; i >= 0; --i) {
        if (a[i][i] == 0.0f) return 0;
        float s = x[i];
        for (int c = i + 1; c < 6; ++c) s -= a[i][c] * x[c];
        x[i] = s / a[i][i];
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* normals.rs                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* compute_normal_pca (normals.rs:158-205) */
static void compute_normal_pca(const float *xyz, const uint64_t *ind, size_t n, float normal[3]) {
    if (n < 3) { normal[0] = 0.0f; normal[1] = 0.0f; normal[2] = 1.0f; return; }
    float c[3] = { 0.0f, 0.0f, 0.0f };
    for (size_t i = 0; i < n; ++i) {
        const float *p = &xyz[3 * ind[i]];
        c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
    }
    float nf = (float)n;
    c[0] /= nf; c[1] /= nf; c[2] /= nf;
    float cov[3][3] = { {0, 0, 0}, {0, 0, 0}, {0, 0, 0} };
    for (size_t i = 0; i < n; ++i) {
        const float *p = &xyz[3 * ind[i]];
        float d[3] = { p[0] - c[0], p[1] - c[1], p[2] - c[2] };
        for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) cov[r][s] += d[r] * d[s];
    }
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) cov[r][s] /= nf;
    float ev[3], q[3][3];
    sym_eigen3(cov, ev, q);
    int mi = 0;
    for (int i = 1; i < 3; ++i) if (ev[i] < ev[mi]) mi = i;
    float nx = q[0][mi], ny = q[1][mi], nz = q[2][mi];
    float mag = sqrtf(nx * nx + ny * ny + nz * nz);
    if (mag > 1e-6f) { normal[0] = nx / mag; normal[1] = ny / mag; normal[2] = nz / mag; }
    else { normal[0] = 0.0f; normal[1] = 0.0f; normal[2] = 1.0f; }
}

/* orient_normal_towards_viewpoint (normals.rs:208-222) */
static void orient_normal(float n[3], const float p[3], const float vp[3]) {
    float d[3] = { vp[0] - p[0], vp[1] - p[1], vp[2] - p[2] };
    float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float t[3] = { d[0] / nrm, d[1] / nrm, d[2] / nrm };
    float dp = n[0] * t[0] + n[1] * t[1] + n[2] * t[2];
    if (dp < 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
}

/* "kNN(k+1) minus self, take k" (normals.rs:147-153 and :316-322) */
static size_t knn_minus_self(const tco_kdtree *t, const float q[3], size_t self_idx, size_t k,
                             kd_scratch *s, uint64_t *out) {
    size_t m = kd_knn_core(t, q, k + 1, s);
    size_t c = 0;
    for (size_t j = 0; j < m && c < k; ++j) if (s->heap[j].idx != self_idx) out[c++] = s->heap[j].idx;
    return c;
}

int tco_estimate_normals(const float *xyz, size_t n, size_t k, float radius, int has_radius,
                         int consistent_orientation, const float *viewpoint, float *out6, int threads) {
    if (n == 0) return TCO_OK;            /* :261-263 (before the k check) */
    if (k < 3) return TCO_INVALID_DATA;   /* :265-269 */
    tco_kdtree *tree = tco_kdtree_new(xyz, n);
    float vp[3];
    if (viewpoint) { vp[0] = viewpoint[0]; vp[1] = viewpoint[1]; vp[2] = viewpoint[2]; }
    else {  /* :275-303 sequential f32 min/max */
        float mnx = xyz[0], mny = xyz[1], mnz = xyz[2], mxx = xyz[0], mxy = xyz[1], mxz = xyz[2];
        for (size_t i = 0; i < n; ++i) {
            mnx = fminf(mnx, xyz[3 * i]); mny = fminf(mny, xyz[3 * i + 1]); mnz = fminf(mnz, xyz[3 * i + 2]);
            mxx = fmaxf(mxx, xyz[3 * i]); mxy = fmaxf(mxy, xyz[3 * i + 1]); mxz = fmaxf(mxz, xyz[3 * i + 2]);
        }
        float cx = (mnx + mxx) / 2.0f, cy = (mny + mxy) / 2.0f, cz = (mnz + mxz) / 2.0f;
        float ex = mxx - mnx, ey = mxy - mny, ez = mxz - mnz;
        float extent = sqrtf(ex * ex + ey * ey + ez * ez);
        vp[0] = cx + 0.0f; vp[1] = cy + 0.0f; vp[2] = cz + extent;
    }
    int nt = resolve_threads(threads);
    (void)nt;
#pragma omp parallel num_threads(nt)
    {
        kd_scratch s; scratch_init(&s);
        size_t cap = (k > 5 ? k : 5) + 2;
        uint64_t *nb = (uint64_t *)malloc(cap * sizeof(uint64_t));
#pragma omp for schedule(dynamic, 512)
        for (long long ii = 0; ii < (long long)n; ++ii) {
            size_t i = (size_t)ii;
            const float *p = &xyz[3 * i];
            size_t cnt = 0;
            if (has_radius) {   /* :141-146 */
                nb_t *res; size_t m = kd_radius_core(tree, p, radius, &res, &s);
                if (m + 1 > cap) { cap = m + 2; nb = (uint64_t *)realloc(nb, cap * sizeof(uint64_t)); }
                for (size_t j = 0; j < m; ++j) if (res[j].idx != i) nb[cnt++] = res[j].idx;
                free(res);
                if (cnt < k) cnt = knn_minus_self(tree, p, i, k, &s, nb);   /* :315-323 */
            } else {
                cnt = knn_minus_self(tree, p, i, k, &s, nb);
            }
            if (cnt < 3) {  /* :326-336 */
                size_t fk = k > 5 ? k : 5;
                cnt = knn_minus_self(tree, p, i, fk, &s, nb);
            }
            int has_self = 0;
            for (size_t j = 0; j < cnt; ++j) if (nb[j] == i) { has_self = 1; break; }
            if (!has_self) nb[cnt++] = i;  /* :338-340 */
            float nrm[3];
            compute_normal_pca(xyz, nb, cnt, nrm);
            if (consistent_orientation) orient_normal(nrm, p, vp);
            float *o = &out6[6 * i];
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = nrm[0]; o[4] = nrm[1]; o[5] = nrm[2];
        }
        free(nb); scratch_free(&s);
    }
    tco_kdtree_free(tree);
    return TCO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* registration.rs                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* find_correspondences_with_tree (registration.rs:87-107); tgt_idx = UINT64_MAX for None */
static void find_correspondences(const float *tsrc, size_t ns, const tco_kdtree *tree, float max_dist,
                                 uint64_t *tgt_idx, int threads) {
    int nt = resolve_threads(threads);
    (void)nt;
#pragma omp parallel num_threads(nt)
    {
        kd_scratch s; scratch_init(&s);
#pragma omp for schedule(dynamic, 512)
        for (long long j = 0; j < (long long)ns; ++j) {
            size_t m = kd_knn_core(tree, &tsrc[3 * j], 1, &s);
            if (m == 0) { tgt_idx[j] = UINT64_MAX; continue; }
            float distance = sqrtf(s.heap[0].d);
            if (max_dist >= 0.0f && distance > max_dist) tgt_idx[j] = UINT64_MAX;
            else tgt_idx[j] = s.heap[0].idx;
        }
        scratch_free(&s);
    }
}

/* DIAGNOSTIC switch, see the comment above p2plane_solve (default 0 = the reference's arithmetic) */
static int g_exact_sums = 0;
void tco_set_exact_sums(int on) { g_exact_sums = on; }

/* compute_transformation (registration.rs:144-203) */
static int kabsch(const float *vs, const float *vq, size_t n, float out[7]) {
    float nf = (float)n;
    float cs[3] = { 0, 0, 0 }, cq[3] = { 0, 0, 0 };
    float h[3][3] = { {0, 0, 0}, {0, 0, 0}, {0, 0, 0} };
    if (g_exact_sums) {          /* the same f32 terms, added in f64 */
        double dcs[3] = { 0, 0, 0 }, dcq[3] = { 0, 0, 0 }, dh[3][3] = { {0, 0, 0}, {0, 0, 0}, {0, 0, 0} };
        for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) { dcs[c] += (double)vs[3 * i + c]; dcq[c] += (double)vq[3 * i + c]; }
        for (int c = 0; c < 3; ++c) { cs[c] = (float)dcs[c] / nf; cq[c] = (float)dcq[c] / nf; }
        for (size_t i = 0; i < n; ++i) {
            float p[3] = { vs[3 * i] - cs[0], vs[3 * i + 1] - cs[1], vs[3 * i + 2] - cs[2] };
            float q[3] = { vq[3 * i] - cq[0], vq[3 * i + 1] - cq[1], vq[3 * i + 2] - cq[2] };
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dh[r][c] += (double)(p[r] * q[c]);
        }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) h[r][c] = (float)dh[r][c];
    } else {
    for (size_t i = 0; i < n; ++i) { cs[0] = cs[0] + vs[3 * i]; cs[1] = cs[1] + vs[3 * i + 1]; cs[2] = cs[2] + vs[3 * i + 2]; }
    cs[0] /= nf; cs[1] /= nf; cs[2] /= nf;
    for (size_t i = 0; i < n; ++i) { cq[0] = cq[0] + vq[3 * i]; cq[1] = cq[1] + vq[3 * i + 1]; cq[2] = cq[2] + vq[3 * i + 2]; }
    cq[0] /= nf; cq[1] /= nf; cq[2] /= nf;
    for (size_t i = 0; i < n; ++i) {
        float p[3] = { vs[3 * i] - cs[0], vs[3 * i + 1] - cs[1], vs[3 * i + 2] - cs[2] };
        float q[3] = { vq[3 * i] - cq[0], vq[3 * i + 1] - cq[1], vq[3 * i + 2] - cq[2] };
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) h[r][c] += p[r] * q[c];
    }
    }
    float U[3][3], w[3], Vt[3][3];
    svd3(h, U, w, Vt);
    float V[3][3], Ut[3][3], R[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { V[r][c] = Vt[c][r]; Ut[r][c] = U[c][r]; }
    mat3_mul(V, Ut, R);
    if (mat3_det(R) < 0.0f) {
        for (int c = 0; c < 3; ++c) Vt[2][c] = -Vt[2][c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) V[r][c] = Vt[c][r];
        mat3_mul(V, Ut, R);
    }
    float q[4]; quat_from_matrix(R, q);
    float rc[3]; quat_rotate(q, cs, rc);
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
    out[4] = cq[0] - rc[0]; out[5] = cq[1] - rc[1]; out[6] = cq[2] - rc[2];
    return 1;
}
/* batch form of tco_symmetric_eigen3 for the differential tests on real covariance matrices (tests/test_oracle_pinning.py) */
void tco_symmetric_eigen3_batch(const float *m9, size_t n, float *evals3, float *evecs9) {
#pragma omp parallel for schedule(static) num_threads(tco_num_threads())
    for (long long i = 0; i < (long long)n; ++i) tco_symmetric_eigen3(&m9[9 * i], &evals3[3 * i], &evecs9[9 * i]);
}

void tco_kabsch(const float *s, const float *q, size_t n, float out7[7], int *ok) { *ok = kabsch(s, q, n, out7); }

/* compute_mse (registration.rs:206-218) */
static float compute_mse(const float *vs, const float *vq, size_t n) {
    if (n == 0) return 0.0f;
    float sum = 0.0f;
    double dsum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        float dx = vs[3 * i] - vq[3 * i], dy = vs[3 * i + 1] - vq[3 * i + 1], dz = vs[3 * i + 2] - vq[3 * i + 2];
        sum += dx * dx + dy * dy + dz * dz;
        dsum += (double)(dx * dx + dy * dy + dz * dz);
    }
    if (g_exact_sums) return (float)(dsum / (double)n);
    return sum / (float)n;
}

static void result_store_corr(tco_icp_result *res, const uint64_t *cs, const uint64_t *ct, size_t n) {
    res->n_corr = n;
    if (res->corr_src && res->corr_tgt) { memcpy(res->corr_src, cs, n * sizeof(uint64_t)); memcpy(res->corr_tgt, ct, n * sizeof(uint64_t)); }
}

int tco_icp_point_to_point(const float *src, size_t ns, const float *tgt, size_t nt,
                           const float init[7], size_t max_iters, float max_dist, float conv_thr,
                           tco_icp_result *res, int threads) {
    if (ns == 0 || nt == 0) return TCO_INVALID_DATA;   /* :266-270 */
    if (max_iters == 0) return TCO_INVALID_DATA;       /* :272-276 */
    float cur[7]; memcpy(cur, init, sizeof(cur));
    float prev_mse = INFINITY;
    tco_kdtree *tree = tco_kdtree_new(tgt, nt);
    float *ts = (float *)malloc(ns * 3 * sizeof(float));
    float *vs = (float *)malloc(ns * 3 * sizeof(float));
    float *vq = (float *)malloc(ns * 3 * sizeof(float));
    uint64_t *ti = (uint64_t *)malloc(ns * sizeof(uint64_t));
    uint64_t *pcs = (uint64_t *)malloc(ns * sizeof(uint64_t)), *pct = (uint64_t *)malloc(ns * sizeof(uint64_t));
    uint64_t *fcs = (uint64_t *)malloc(ns * sizeof(uint64_t)), *fct = (uint64_t *)malloc(ns * sizeof(uint64_t));
    size_t nfinal = 0;
    int rc = TCO_OK, done = 0;
    for (size_t it = 0; it < max_iters && !done; ++it) {
        for (size_t j = 0; j < ns; ++j) tco_isometry_apply(cur, &src[3 * j], &ts[3 * j]);   /* :285-289 */
        find_correspondences(ts, ns, tree, max_dist, ti, threads);                          /* :292-296 */
        size_t nv = 0;
        for (size_t j = 0; j < ns; ++j) if (ti[j] != UINT64_MAX) {                          /* :299-309 */
            memcpy(&vs[3 * nv], &ts[3 * j], 12); memcpy(&vq[3 * nv], &tgt[3 * ti[j]], 12);
            pcs[nv] = j; pct[nv] = ti[j]; ++nv;
        }
        if (nv < 3) { rc = TCO_ALGORITHM; break; }                                          /* :311-315 */
        float delta[7], nxt[7];
        kabsch(vs, vq, nv, delta);
        tco_isometry_mul(delta, cur, nxt); memcpy(cur, nxt, sizeof(cur));                   /* :321 */
        float mse = compute_mse(vs, vq, nv);                                                /* :324 */
        float change = fabsf(prev_mse - mse);
        if (change < conv_thr) {                                                            /* :327-336 */
            memcpy(res->transform, cur, sizeof(cur)); res->mse = mse; res->iterations = it + 1; res->converged = 1;
            result_store_corr(res, pcs, pct, nv);
            done = 1; break;
        }
        prev_mse = mse;
        uint64_t *t1 = fcs; fcs = pcs; pcs = t1; t1 = fct; fct = pct; pct = t1; nfinal = nv;
    }
    if (rc == TCO_OK && !done) {   /* :343-369 */
        for (size_t j = 0; j < ns; ++j) tco_isometry_apply(cur, &src[3 * j], &ts[3 * j]);
        float fm;
        if (nfinal) {
            for (size_t i = 0; i < nfinal; ++i) { memcpy(&vs[3 * i], &ts[3 * fcs[i]], 12); memcpy(&vq[3 * i], &tgt[3 * fct[i]], 12); }
            fm = compute_mse(vs, vq, nfinal);
        } else fm = prev_mse;
        memcpy(res->transform, cur, sizeof(cur)); res->mse = fm; res->iterations = max_iters; res->converged = 0;
        result_store_corr(res, fcs, fct, nfinal);
    }
    free(ts); free(vs); free(vq); free(ti); free(pcs); free(pct); free(fcs); free(fct);
    tco_kdtree_free(tree);
    return rc;
}

int tco_icp_point_to_point_checked(const float *src, size_t ns, const float *tgt, size_t nt,
                           const float init[7], size_t max_iters, float conv_thr, float max_dist,
                           tco_icp_result *res, int threads) {
    if (ns == 0 || nt == 0) return TCO_INVALID_DATA;   /* :653-657 */
    if (max_iters == 0) return TCO_INVALID_DATA;       /* :659-663 */
    if (conv_thr <= 0.0f) return TCO_INVALID_DATA;     /* :665-669 */
    return tco_icp_point_to_point(src, ns, tgt, nt, init, max_iters, max_dist, conv_thr, res, threads);
}

void tco_icp(const float *src, size_t ns, const float *tgt, size_t nt,
             const float init[7], size_t max_iters, float out[7], int threads) {
    tco_icp_result r; memset(&r, 0, sizeof(r));
    int rc = tco_icp_point_to_point(src, ns, tgt, nt, init, max_iters, -1.0f, 1e-6f, &r, threads); /* :238 */
    if (rc == TCO_OK) memcpy(out, r.transform, 7 * sizeof(float));
    else memcpy(out, init, 7 * sizeof(float));   /* :240 */
}

/* DIAGNOSTIC switch (default 0 = the reference's arithmetic).  The reference adds the per-pair terms of its 6x6 system one
 * after the other in f32 (registration.rs:409-428): with ~10^6 pairs every sum carries a rounding error of ~1e-5 relative
 * (terms below half an ulp of the running sum are even dropped outright).  With the switch on, the SAME f32 per-pair terms are
 * added in f64, i.e. the sums the reference's formula defines, without its accumulation error.  Tests use it to show that a
 * transform which is 2e-5 away from the reference's on a 10^6-point surface is that far away because of the reference's own
 * accumulation error, not because of different pairs or a different solve (tests/test_gpu_fullsize.py).
 * The switch (defined above compute_transformation) also covers the Kabsch sums and both mse sums, i.e. every sum over the
 * pairs of an iteration: tools/dev/loop_fuzz.py uses it to tell a stop decision taken within the reference's own rounding
 * (|prev_mse - mse| < threshold) from a defect. */
/* compute_transformation_point_to_plane (registration.rs:395-450) */
static int p2plane_solve(const float *vs, const float *vq, const float *vn, size_t n, float out[7]) {
    float ata[36]; float atb[6];
    memset(ata, 0, sizeof(ata)); memset(atb, 0, sizeof(atb));
    double data[36], datb[6];
    memset(data, 0, sizeof(data)); memset(datb, 0, sizeof(datb));
    for (size_t i = 0; i < n; ++i) {
        const float *s = &vs[3 * i], *q = &vq[3 * i], *nn = &vn[3 * i];
        float c[3]; cross3(s, nn, c);
        float a[6] = { c[0], c[1], c[2], nn[0], nn[1], nn[2] };
        float d[3] = { q[0] - s[0], q[1] - s[1], q[2] - s[2] };
        float b = nn[0] * d[0] + nn[1] * d[1] + nn[2] * d[2];
        if (g_exact_sums) {
            for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) data[6 * r + cc] += (double)(a[r] * a[cc]);
            for (int r = 0; r < 6; ++r) datb[r] += (double)(a[r] * b);
            continue;
        }
        for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 6; ++cc) ata[6 * r + cc] += a[r] * a[cc];
        for (int r = 0; r < 6; ++r) atb[r] += a[r] * b;
    }
    if (g_exact_sums) { for (int e = 0; e < 36; ++e) ata[e] = (float)data[e]; for (int e = 0; e < 6; ++e) atb[e] = (float)datb[e]; }
    float x[6];
    if (!tco_cholesky6_solve(ata, atb, x)) { if (!tco_lu6_solve(ata, atb, x)) return 0; }   /* :432-438 */
    /* rot = Rz(x2) * Ry(x1) * Rx(x0), axis-angle unit quaternions (:441-444) */
    float hx = x[0] / 2.0f, hy = x[1] / 2.0f, hz = x[2] / 2.0f;
    float qx[4] = { 1.0f * sinf(hx), 0.0f * sinf(hx), 0.0f * sinf(hx), cosf(hx) };
    float qy[4] = { 0.0f * sinf(hy), 1.0f * sinf(hy), 0.0f * sinf(hy), cosf(hy) };
    float qz[4] = { 0.0f * sinf(hz), 0.0f * sinf(hz), 1.0f * sinf(hz), cosf(hz) };
    float zy[4], rot[4];
    quat_mul(qz, qy, zy); quat_mul(zy, qx, rot);
    out[0] = rot[0]; out[1] = rot[1]; out[2] = rot[2]; out[3] = rot[3];
    out[4] = x[3]; out[5] = x[4]; out[6] = x[5];
    return 1;
}

/* compute_point_to_plane_mse (registration.rs:453-471) */
static float p2plane_mse(const float *vs, const float *vq, const float *vn, size_t n) {
    if (n == 0) return 0.0f;
    float sum = 0.0f;
    double dsum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        float d[3] = { vq[3 * i] - vs[3 * i], vq[3 * i + 1] - vs[3 * i + 1], vq[3 * i + 2] - vs[3 * i + 2] };
        float dd = vn[3 * i] * d[0] + vn[3 * i + 1] * d[1] + vn[3 * i + 2] * d[2];
        sum += dd * dd;
        dsum += (double)(dd * dd);
    }
    if (g_exact_sums) return (float)(dsum / (double)n);
    return sum / (float)n;
}

int tco_icp_point_to_plane(const float *src, size_t ns, const float *tgt, size_t nt,
                           const float *tgt_normals, size_t n_normals,
                           const float init[7], size_t max_iters, float max_dist, float conv_thr,
                           tco_icp_result *res, int threads) {
    if (ns == 0 || nt == 0) return TCO_INVALID_DATA;    /* :517-521 */
    if (n_normals != nt) return TCO_INVALID_DATA;       /* :522-526 */
    if (max_iters == 0) return TCO_INVALID_DATA;        /* :527-531 */
    float cur[7]; memcpy(cur, init, sizeof(cur));
    float prev_mse = INFINITY;
    tco_kdtree *tree = tco_kdtree_new(tgt, nt);
    float *ts = (float *)malloc(ns * 3 * sizeof(float));
    float *vs = (float *)malloc(ns * 3 * sizeof(float));
    float *vq = (float *)malloc(ns * 3 * sizeof(float));
    float *vn = (float *)malloc(ns * 3 * sizeof(float));
    uint64_t *ti = (uint64_t *)malloc(ns * sizeof(uint64_t));
    uint64_t *pcs = (uint64_t *)malloc(ns * sizeof(uint64_t)), *pct = (uint64_t *)malloc(ns * sizeof(uint64_t));
    uint64_t *fcs = (uint64_t *)malloc(ns * sizeof(uint64_t)), *fct = (uint64_t *)malloc(ns * sizeof(uint64_t));
    size_t nfinal = 0;
    int rc = TCO_OK, done = 0;
    for (size_t it = 0; it < max_iters; ++it) {
        for (size_t j = 0; j < ns; ++j) tco_isometry_apply(cur, &src[3 * j], &ts[3 * j]);   /* :540-544 */
        find_correspondences(ts, ns, tree, max_dist, ti, threads);                          /* :547-551 */
        size_t nv = 0;
        for (size_t j = 0; j < ns; ++j) if (ti[j] != UINT64_MAX) {                          /* :558-565 */
            memcpy(&vs[3 * nv], &ts[3 * j], 12); memcpy(&vq[3 * nv], &tgt[3 * ti[j]], 12);
            memcpy(&vn[3 * nv], &tgt_normals[3 * ti[j]], 12);
            pcs[nv] = j; pct[nv] = ti[j]; ++nv;
        }
        if (nv < 6) { rc = TCO_ALGORITHM; break; }                                          /* :568-572 */
        float delta[7], nxt[7];
        if (!p2plane_solve(vs, vq, vn, nv, delta)) { rc = TCO_ALGORITHM; break; }
        tco_isometry_mul(delta, cur, nxt); memcpy(cur, nxt, sizeof(cur));                   /* :576 */
        float mse = p2plane_mse(vs, vq, vn, nv);                                            /* :578 */
        float change = fabsf(prev_mse - mse);
        if (change < conv_thr) {                                                            /* :581-589 */
            memcpy(res->transform, cur, sizeof(cur)); res->mse = mse; res->iterations = it + 1; res->converged = 1;
            result_store_corr(res, pcs, pct, nv);
            done = 1; break;
        }
        prev_mse = mse;
        uint64_t *t1 = fcs; fcs = pcs; pcs = t1; t1 = fct; fct = pct; pct = t1; nfinal = nv;
    }
    if (rc == TCO_OK && !done) {   /* :595-601: previous_mse, no recompute */
        memcpy(res->transform, cur, sizeof(cur)); res->mse = prev_mse; res->iterations = max_iters; res->converged = 0;
        result_store_corr(res, fcs, fct, nfinal);
    }
    free(ts); free(vs); free(vq); free(vn); free(ti); free(pcs); free(pct); free(fcs); free(fct);
    tco_kdtree_free(tree);
    return rc;
}

/* ---- per-iteration packed sums in f64 (building blocks for the sharded / gloo tests) ---- */
int tco_p2plane_partial(const float *src, size_t j0, size_t j1, const tco_kdtree *tree,
                        const float *tgt, const float *tgt_normals, const float T[7],
                        float max_dist, double out[29], uint32_t *corr) {
    for (int i = 0; i < 29; ++i) out[i] = 0.0;
    kd_scratch s; scratch_init(&s);
    for (size_t j = j0; j < j1; ++j) {
        float ts[3]; tco_isometry_apply(T, &src[3 * j], ts);
        size_t m = kd_knn_core(tree, ts, 1, &s);
        uint32_t ci = 0xFFFFFFFFu;
        if (m) {
            float distance = sqrtf(s.heap[0].d);
            if (!(max_dist >= 0.0f && distance > max_dist)) ci = (uint32_t)s.heap[0].idx;
        }
        if (corr) corr[j - j0] = ci;
        if (ci == 0xFFFFFFFFu) continue;
        const float *q = &tgt[3 * (size_t)ci], *nn = &tgt_normals[3 * (size_t)ci];
        float c[3]; cross3(ts, nn, c);
        float a[6] = { c[0], c[1], c[2], nn[0], nn[1], nn[2] };
        float d[3] = { q[0] - ts[0], q[1] - ts[1], q[2] - ts[2] };
        float b = nn[0] * d[0] + nn[1] * d[1] + nn[2] * d[2];
        int o = 0;
        for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc) out[o++] += (double)a[r] * (double)a[cc];
        for (int r = 0; r < 6; ++r) out[21 + r] += (double)a[r] * (double)b;
        out[27] += (double)b * (double)b;
        out[28] += 1.0;
    }
    scratch_free(&s);
    return TCO_OK;
}

int tco_p2p_partial(const float *src, size_t j0, size_t j1, const tco_kdtree *tree,
                    const float *tgt, const float T[7], float max_dist, double out[17], uint32_t *corr) {
    for (int i = 0; i < 17; ++i) out[i] = 0.0;
    kd_scratch s; scratch_init(&s);
    for (size_t j = j0; j < j1; ++j) {
        float ts[3]; tco_isometry_apply(T, &src[3 * j], ts);
        size_t m = kd_knn_core(tree, ts, 1, &s);
        uint32_t ci = 0xFFFFFFFFu;
        if (m) {
            float distance = sqrtf(s.heap[0].d);
            if (!(max_dist >= 0.0f && distance > max_dist)) ci = (uint32_t)s.heap[0].idx;
        }
        if (corr) corr[j - j0] = ci;
        if (ci == 0xFFFFFFFFu) continue;
        const float *q = &tgt[3 * (size_t)ci];
        for (int r = 0; r < 3; ++r) { out[r] += (double)ts[r]; out[3 + r] += (double)q[r]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out[6 + 3 * r + c] += (double)ts[r] * (double)q[c];
        float dx = ts[0] - q[0], dy = ts[1] - q[1], dz = ts[2] - q[2];
        out[15] += (double)(dx * dx + dy * dy + dz * dz);
        out[16] += 1.0;
    }
    scratch_free(&s);
    return TCO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* filtering.rs:38-133 voxel_grid_filter                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t k[3]; double s[3]; size_t cnt; } vox_t;
static int cmp_vox(const void *a, const void *b) {
    const vox_t *x = (const vox_t *)a, *y = (const vox_t *)b;
    for (int i = 0; i < 3; ++i) { if (x->k[i] < y->k[i]) return -1; if (x->k[i] > y->k[i]) return 1; }
    return 0;
}
static int cmp_tb(const void *a, const void *b) {
    int c = cmp_vox(a, b);
    if (c) return c;
    size_t x = ((const vox_t *)a)->cnt, y = ((const vox_t *)b)->cnt;
    return (x < y) ? -1 : (x > y) ? 1 : 0;
}
int tco_voxel_grid_filter(const float *xyz, size_t n, float voxel, float *out, size_t *n_out) {
    *n_out = 0;
    if (n == 0) return TCO_OK;                 /* :42-44 */
    if (voxel <= 0.0f) return TCO_INVALID_DATA; /* :46-50 */
    float mn[3] = { xyz[0], xyz[1], xyz[2] };
    for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) if (xyz[3 * i + c] < mn[c]) mn[c] = xyz[3 * i + c];
    vox_t *v = (vox_t *)malloc(n * sizeof(vox_t));
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) {
            v[i].k[c] = (int32_t)floorf((xyz[3 * i + c] - mn[c]) / voxel);   /* :96-101 */
            v[i].s[c] = (double)xyz[3 * i + c];
        }
        v[i].cnt = i;   /* original order, to keep the per-voxel f64 sum in input order */
    }
    /* group by key; ties keep input order so each voxel's f64 sum runs in input order */
    qsort(v, n, sizeof(vox_t), cmp_tb);
    size_t o = 0, i = 0;
    while (i < n) {
        double s[3] = { 0.0, 0.0, 0.0 }; size_t c = 0; size_t j = i;
        while (j < n && cmp_vox(&v[i], &v[j]) == 0) { s[0] += v[j].s[0]; s[1] += v[j].s[1]; s[2] += v[j].s[2]; ++c; ++j; }
        double inv = 1.0 / (double)c;   /* :122-128 */
        out[3 * o] = (float)(s[0] * inv); out[3 * o + 1] = (float)(s[1] * inv); out[3 * o + 2] = (float)(s[2] * inv);
        ++o; i = j;
    }
    free(v);
    *n_out = o;
    return TCO_OK;
}


/* ---- KISS-ICP (threecrate-algorithms/src/kiss_icp.rs) ---------------------------------------
 * range_filter :56-70, adaptive_threshold :82-95, svd_transform :102-162 (= compute_transformation
 * plus the |H|_F < 1e-10 rejection), kiss_icp :183-300 (mse AFTER applying delta, fixed 1e-6 rule,
 * not converged -> prev_mse).  The downsampled source comes from tco_voxel_grid_filter, whose output
 * order (sorted by voxel key) stands in for the reference's unspecified HashMap order: correspondence
 * source indices refer to that order. */
static int kiss_svd_transform(const float *vs, const float *vq, size_t n, float out[7]) {
    if (n < 3) return TCO_ALGORITHM;
    float nf = (float)n;
    float cs[3] = { 0, 0, 0 }, cq[3] = { 0, 0, 0 };
    float h[3][3] = { {0, 0, 0}, {0, 0, 0}, {0, 0, 0} };
    if (g_exact_sums) {          /* diagnostic (tco_set_exact_sums): the same f32 terms, added in f64 */
        double dcs[3] = { 0, 0, 0 }, dcq[3] = { 0, 0, 0 }, dh[3][3] = { {0, 0, 0}, {0, 0, 0}, {0, 0, 0} };
        for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) { dcs[c] += (double)vs[3 * i + c]; dcq[c] += (double)vq[3 * i + c]; }
        for (int c = 0; c < 3; ++c) { cs[c] = (float)dcs[c] / nf; cq[c] = (float)dcq[c] / nf; }
        for (size_t i = 0; i < n; ++i) {
            float p[3] = { vs[3 * i] - cs[0], vs[3 * i + 1] - cs[1], vs[3 * i + 2] - cs[2] };
            float q[3] = { vq[3 * i] - cq[0], vq[3 * i + 1] - cq[1], vq[3 * i + 2] - cq[2] };
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dh[r][c] += (double)(p[r] * q[c]);
        }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) h[r][c] = (float)dh[r][c];
    } else {
    for (size_t i = 0; i < n; ++i) { cs[0] = cs[0] + vs[3 * i]; cs[1] = cs[1] + vs[3 * i + 1]; cs[2] = cs[2] + vs[3 * i + 2]; }
    cs[0] /= nf; cs[1] /= nf; cs[2] /= nf;
    for (size_t i = 0; i < n; ++i) { cq[0] = cq[0] + vq[3 * i]; cq[1] = cq[1] + vq[3 * i + 1]; cq[2] = cq[2] + vq[3 * i + 2]; }
    cq[0] /= nf; cq[1] /= nf; cq[2] /= nf;
    for (size_t i = 0; i < n; ++i) {
        float p[3] = { vs[3 * i] - cs[0], vs[3 * i + 1] - cs[1], vs[3 * i + 2] - cs[2] };
        float q[3] = { vq[3 * i] - cq[0], vq[3 * i + 1] - cq[1], vq[3 * i + 2] - cq[2] };
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) h[r][c] += p[r] * q[c];
    }
    }
    float hn = 0.0f;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) hn += h[r][c] * h[r][c];
    if (sqrtf(hn) < 1e-10f) return TCO_ALGORITHM;                       /* :130-136 */
    float U[3][3], w[3], Vt[3][3];
    svd3(h, U, w, Vt);
    float V[3][3], Ut[3][3], R[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { V[r][c] = Vt[c][r]; Ut[r][c] = U[c][r]; }
    mat3_mul(V, Ut, R);
    if (mat3_det(R) < 0.0f) {
        for (int c = 0; c < 3; ++c) Vt[2][c] = -Vt[2][c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) V[r][c] = Vt[c][r];
        mat3_mul(V, Ut, R);
    }
    float q[4]; quat_from_matrix(R, q);
    float rc[3]; quat_rotate(q, cs, rc);
    out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
    out[4] = cq[0] - rc[0]; out[5] = cq[1] - rc[1]; out[6] = cq[2] - rc[2];
    return TCO_OK;
}

float tco_kiss_adaptive_threshold(const float init[7], float voxel_size) {      /* :82-95 */
    float trans = sqrtf(init[4] * init[4] + init[5] * init[5] + init[6] * init[6]);
    float imag = sqrtf(init[0] * init[0] + init[1] * init[1] + init[2] * init[2]);
    float rot_disp = 2.0f * imag * voxel_size;
    float motion = trans + rot_disp;
    return fminf(fmaxf(3.0f * motion, 3.0f * voxel_size), 10.0f * voxel_size);
}

/* The reference's voxel_grid_filter emits its voxels in HashMap iteration order (filtering.rs:120-130), which Rust randomises
 * per process: kiss_icp's sequential f32 sums over the down-sampled source therefore run in an UNSPECIFIED order.  The oracle
 * emits voxels sorted by key; a non-zero seed applies a deterministic shuffle to the down-sampled source instead, so that
 * tests can measure how far the reference's own result moves with that order (tests/test_gpu_parity.py). */
static uint64_t g_voxel_order_seed = 0;
void tco_set_voxel_order_seed(uint64_t seed) { g_voxel_order_seed = seed; }

int tco_kiss_icp(const float *src, size_t ns, const float *tgt, size_t nt, const float init[7],
                 float voxel_size, float max_range, float min_range, size_t max_iters,
                 tco_icp_result *res, size_t *n_source_down, int threads) {
    if (ns == 0 || nt == 0) return TCO_INVALID_DATA;        /* :189-193 */
    if (max_iters == 0) return TCO_INVALID_DATA;            /* :194-198 */
    if (voxel_size <= 0.0f) return TCO_INVALID_DATA;        /* :199-203 */
    float min_sq = min_range * min_range, max_sq = max_range * max_range;
    float *ranged = (float *)malloc(ns * 3 * sizeof(float));
    size_t nr = 0;
    for (size_t j = 0; j < ns; ++j) {                        /* range_filter :56-70 */
        const float *p = &src[3 * j];
        float r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        if (r2 >= min_sq && r2 <= max_sq) { memcpy(&ranged[3 * nr], p, 12); ++nr; }
    }
    if (nr == 0) { free(ranged); return TCO_INVALID_DATA; }  /* :207-213 */
    float *down = (float *)malloc(nr * 3 * sizeof(float));
    size_t nd = 0;
    int vrc = tco_voxel_grid_filter(ranged, nr, voxel_size, down, &nd);
    free(ranged);
    if (vrc != TCO_OK || nd == 0) { free(down); return TCO_INVALID_DATA; }
    if (n_source_down) *n_source_down = nd;
    if (g_voxel_order_seed) {                                /* Fisher-Yates over the voxels, SplitMix64 */
        uint64_t z = g_voxel_order_seed;
        for (size_t i = nd - 1; i > 0; --i) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t r = z; r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull; r = (r ^ (r >> 27)) * 0x94D049BB133111EBull; r ^= r >> 31;
            size_t k = (size_t)(r % (i + 1));
            float t3[3]; memcpy(t3, &down[3 * i], 12); memcpy(&down[3 * i], &down[3 * k], 12); memcpy(&down[3 * k], t3, 12);
        }
    }
    float sigma = tco_kiss_adaptive_threshold(init, voxel_size);
    tco_kdtree *tree = tco_kdtree_new(tgt, nt);
    float cur[7]; memcpy(cur, init, sizeof(cur));
    float prev_mse = INFINITY;
    float *ts = (float *)malloc(nd * 3 * sizeof(float));
    float *vs = (float *)malloc(nd * 3 * sizeof(float));
    float *vq = (float *)malloc(nd * 3 * sizeof(float));
    uint64_t *ti = (uint64_t *)malloc(nd * sizeof(uint64_t));
    uint64_t *pcs = (uint64_t *)malloc(nd * sizeof(uint64_t)), *pct = (uint64_t *)malloc(nd * sizeof(uint64_t));
    uint64_t *fcs = (uint64_t *)malloc(nd * sizeof(uint64_t)), *fct = (uint64_t *)malloc(nd * sizeof(uint64_t));
    size_t nfinal = 0;
    int rc = TCO_OK, done = 0;
    for (size_t it = 0; it < max_iters && !done; ++it) {
        for (size_t j = 0; j < nd; ++j) tco_isometry_apply(cur, &down[3 * j], &ts[3 * j]);
        find_correspondences(ts, nd, tree, sigma, ti, threads);          /* dist > sigma -> skip :244-250 */
        size_t nv = 0;
        for (size_t j = 0; j < nd; ++j) if (ti[j] != UINT64_MAX) {
            memcpy(&vs[3 * nv], &ts[3 * j], 12); memcpy(&vq[3 * nv], &tgt[3 * ti[j]], 12);
            pcs[nv] = j; pct[nv] = ti[j]; ++nv;
        }
        if (nv < 3) { rc = TCO_ALGORITHM; break; }                       /* :256-262 */
        float delta[7], nxt[7];
        rc = kiss_svd_transform(vs, vq, nv, delta);
        if (rc != TCO_OK) break;
        tco_isometry_mul(delta, cur, nxt); memcpy(cur, nxt, sizeof(cur));
        float sum = 0.0f;                                                /* mse after delta :270-276 */
        for (size_t i = 0; i < nv; ++i) {
            float d[3]; tco_isometry_apply(delta, &vs[3 * i], d);
            float dx = d[0] - vq[3 * i], dy = d[1] - vq[3 * i + 1], dz = d[2] - vq[3 * i + 2];
            sum += dx * dx + dy * dy + dz * dz;
        }
        float mse = sum / (float)nv;
        if (fabsf(prev_mse - mse) < 1e-6f) {                             /* :278-286 */
            memcpy(res->transform, cur, sizeof(cur)); res->mse = mse; res->iterations = it + 1; res->converged = 1;
            result_store_corr(res, pcs, pct, nv);
            done = 1; break;
        }
        prev_mse = mse;
        uint64_t *t1 = fcs; fcs = pcs; pcs = t1; t1 = fct; fct = pct; pct = t1; nfinal = nv;
    }
    if (rc == TCO_OK && !done) {                                         /* :292-299 */
        memcpy(res->transform, cur, sizeof(cur)); res->mse = prev_mse; res->iterations = max_iters; res->converged = 0;
        result_store_corr(res, fcs, fct, nfinal);
    }
    free(down); free(ts); free(vs); free(vq); free(ti); free(pcs); free(pct); free(fcs); free(fct);
    tco_kdtree_free(tree);
    return rc;
}


/* ---- GICP (threecrate-algorithms/src/gicp.rs) ------------------------------------------------
 * compute_covariances :52-86 (k = max(k, 4) nearest INCLUDING the point itself, f32 mean / outer products
 * in neighbour order, / max(n-1, 1), + 1e-4 I; fewer than 3 neighbours -> 1e-3 I), gicp :100-305
 * (validation, M = C_t + R C_s R^T, nalgebra's closed-form 3x3 try_inverse, sequential f32 sums of the
 * 6x6 H and g, mse = mean dist^2 before the update, Cholesky then LU, Rz Ry Rx update). */
static void gicp_covariances(const float *xyz, size_t n, size_t k, float *cov /* n x 9 row-major */, int threads) {
    if (k < 4) k = 4;
    tco_kdtree *tree = tco_kdtree_new(xyz, n);
    int nt = resolve_threads(threads);
    (void)nt;
#pragma omp parallel num_threads(nt)
    {
        kd_scratch s; scratch_init(&s);
#pragma omp for schedule(dynamic, 256)
        for (long long i = 0; i < (long long)n; ++i) {
            float *c = &cov[9 * i];
            size_t m = kd_knn_core(tree, &xyz[3 * i], k, &s);      /* ascending distance */
            if (m < 3) { for (int e = 0; e < 9; ++e) c[e] = 0.0f; c[0] = c[4] = c[8] = 1.0f * 1e-3f; continue; }
            float nf = (float)m;
            float mean[3] = { 0, 0, 0 };
            for (size_t j = 0; j < m; ++j) { const float *p = &xyz[3 * s.heap[j].idx]; mean[0] = mean[0] + p[0]; mean[1] = mean[1] + p[1]; mean[2] = mean[2] + p[2]; }
            mean[0] /= nf; mean[1] /= nf; mean[2] /= nf;
            float a[9] = { 0 };
            for (size_t j = 0; j < m; ++j) {
                const float *p = &xyz[3 * s.heap[j].idx];
                float d[3] = { p[0] - mean[0], p[1] - mean[1], p[2] - mean[2] };
                for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) a[3 * r + cc] += d[r] * d[cc];
            }
            float den = fmaxf(nf - 1.0f, 1.0f);
            for (int e = 0; e < 9; ++e) c[e] = a[e] / den;
            c[0] += 1e-4f; c[4] += 1e-4f; c[8] += 1e-4f;
        }
        scratch_free(&s);
    }
    tco_kdtree_free(tree);
}

void tco_gicp_covariances(const float *xyz, size_t n, size_t k, float *cov9, int threads) { gicp_covariances(xyz, n, k, cov9, threads); }

/* nalgebra Matrix3::try_inverse (linalg/inverse.rs, 3x3 case): adjugate / determinant, 0 -> None */
static int inv3(const float m[9], float o[9]) {
    float m11 = m[0], m12 = m[1], m13 = m[2], m21 = m[3], m22 = m[4], m23 = m[5], m31 = m[6], m32 = m[7], m33 = m[8];
    float minor_m12_m23 = m22 * m33 - m32 * m23;
    float minor_m11_m23 = m21 * m33 - m31 * m23;
    float minor_m11_m22 = m21 * m32 - m31 * m22;
    float det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
    if (det == 0.0f) return 0;
    o[0] = minor_m12_m23 / det; o[1] = (m13 * m32 - m33 * m12) / det; o[2] = (m12 * m23 - m22 * m13) / det;
    o[3] = -minor_m11_m23 / det; o[4] = (m11 * m33 - m31 * m13) / det; o[5] = (m13 * m21 - m23 * m11) / det;
    o[6] = minor_m11_m22 / det; o[7] = (m12 * m31 - m32 * m11) / det; o[8] = (m11 * m22 - m21 * m12) / det;
    return 1;
}

static void mat3_mul_flat(const float a[9], const float b[9], float c[9]) {
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
        float v = 0.0f;
        for (int k = 0; k < 3; ++k) v += a[3 * r + k] * b[3 * k + cc];
        c[3 * r + cc] = v;
    }
}

int tco_gicp(const float *src, size_t ns, const float *tgt, size_t nt, const float init[7],
             size_t max_iters, float max_dist, float conv_thr, size_t k_corr,
             tco_icp_result *res, int threads) {
    if (ns == 0 || nt == 0) return TCO_INVALID_DATA;                       /* :107-111 */
    if (max_iters == 0) return TCO_INVALID_DATA;                           /* :112-116 */
    size_t min_k = k_corr < 4 ? 4 : k_corr;
    if (ns < min_k || nt < min_k) return TCO_INVALID_DATA;                 /* :120-131 */
    const float *clouds[2] = { src, tgt }; size_t sizes[2] = { ns, nt };
    for (int c = 0; c < 2; ++c) {                                           /* :135-155 */
        float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t i = 0; i < sizes[c]; ++i) for (int ax = 0; ax < 3; ++ax) {
            mn[ax] = fminf(mn[ax], clouds[c][3 * i + ax]); mx[ax] = fmaxf(mx[ax], clouds[c][3 * i + ax]);
        }
        float me = INFINITY;
        for (int ax = 0; ax < 3; ++ax) me = fminf(me, mx[ax] - mn[ax]);
        if (me < 1e-4f) return TCO_INVALID_DATA;
    }
    float *cs = (float *)malloc(ns * 9 * sizeof(float)), *ct = (float *)malloc(nt * 9 * sizeof(float));
    gicp_covariances(src, ns, k_corr, cs, threads);
    gicp_covariances(tgt, nt, k_corr, ct, threads);
    tco_kdtree *tree = tco_kdtree_new(tgt, nt);
    float cur[7]; memcpy(cur, init, sizeof(cur));
    float prev_mse = INFINITY;
    float *ts = (float *)malloc(ns * 3 * sizeof(float));
    uint64_t *ti = (uint64_t *)malloc(ns * sizeof(uint64_t));
    float *td = (float *)malloc(ns * sizeof(float));
    uint64_t *pcs = (uint64_t *)malloc(ns * sizeof(uint64_t)), *pct = (uint64_t *)malloc(ns * sizeof(uint64_t));
    uint64_t *fcs = (uint64_t *)malloc(ns * sizeof(uint64_t)), *fct = (uint64_t *)malloc(ns * sizeof(uint64_t));
    size_t nfinal = 0;
    int rc = TCO_OK, done = 0;
    for (size_t it = 0; it < max_iters; ++it) {
        for (size_t j = 0; j < ns; ++j) tco_isometry_apply(cur, &src[3 * j], &ts[3 * j]);
        float m16[16]; tco_isometry_to_matrix(cur, m16);
        float R[9] = { m16[0], m16[1], m16[2], m16[4], m16[5], m16[6], m16[8], m16[9], m16[10] };
        float Rt[9] = { R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8] };
        /* 1-NN and its distance (parallel), then the sequential accumulation in source order */
        int nthr = resolve_threads(threads);
        (void)nthr;
#pragma omp parallel num_threads(nthr)
        {
            kd_scratch s; scratch_init(&s);
#pragma omp for schedule(dynamic, 512)
            for (long long j = 0; j < (long long)ns; ++j) {
                size_t m = kd_knn_core(tree, &ts[3 * j], 1, &s);
                if (m == 0) { ti[j] = UINT64_MAX; continue; }
                td[j] = sqrtf(s.heap[0].d);
                ti[j] = (td[j] > max_dist) ? UINT64_MAX : s.heap[0].idx;     /* :211-213 */
            }
            scratch_free(&s);
        }
        float H[36]; float g[6];
        memset(H, 0, sizeof(H)); memset(g, 0, sizeof(g));
        double dH[36], dg[6], dmse = 0.0;            /* diagnostic (tco_set_exact_sums): the same f32 terms, added in f64 */
        memset(dH, 0, sizeof(dH)); memset(dg, 0, sizeof(dg));
        size_t n_corr = 0; float mse_sum = 0.0f;
        for (size_t j = 0; j < ns; ++j) {
            if (ti[j] == UINT64_MAX) continue;
            const float *p = &ts[3 * j];
            float tmp[9], rcr[9], M[9], Mi[9];
            mat3_mul_flat(R, &cs[9 * j], tmp); mat3_mul_flat(tmp, Rt, rcr);
            for (int e = 0; e < 9; ++e) M[e] = ct[9 * ti[j] + e] + rcr[e];    /* :217 */
            if (!inv3(M, Mi)) continue;                                       /* :218-221 */
            float r[3] = { tgt[3 * ti[j]] - p[0], tgt[3 * ti[j] + 1] - p[1], tgt[3 * ti[j] + 2] - p[2] };
            /* a = -skew(ts) */
            float a[9] = { -0.0f, p[2], -p[1], -p[2], -0.0f, p[0], p[1], -p[0], -0.0f };
            float at[9] = { a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8] };
            float mia[9], hrr[9], hrt[9];
            mat3_mul_flat(Mi, a, mia); mat3_mul_flat(at, mia, hrr); mat3_mul_flat(at, Mi, hrt);
            float wr[3], gr[3];
            for (int i = 0; i < 3; ++i) { float v = 0.0f; for (int k = 0; k < 3; ++k) v += Mi[3 * i + k] * r[k]; wr[i] = v; }
            for (int i = 0; i < 3; ++i) { float v = 0.0f; for (int k = 0; k < 3; ++k) v += at[3 * i + k] * wr[k]; gr[i] = v; }
            for (int i = 0; i < 3; ++i) {
                for (int jj = 0; jj < 3; ++jj) {
                    H[6 * i + jj] += hrr[3 * i + jj];
                    H[6 * i + jj + 3] += hrt[3 * i + jj];
                    H[6 * (i + 3) + jj] += hrt[3 * jj + i];
                    H[6 * (i + 3) + jj + 3] += Mi[3 * i + jj];
                    dH[6 * i + jj] += (double)hrr[3 * i + jj];
                    dH[6 * i + jj + 3] += (double)hrt[3 * i + jj];
                    dH[6 * (i + 3) + jj] += (double)hrt[3 * jj + i];
                    dH[6 * (i + 3) + jj + 3] += (double)Mi[3 * i + jj];
                }
                g[i] += gr[i];
                g[i + 3] += wr[i];
                dg[i] += (double)gr[i];
                dg[i + 3] += (double)wr[i];
            }
            pcs[n_corr] = j; pct[n_corr] = ti[j];
            n_corr += 1;
            mse_sum += td[j] * td[j];
            dmse += (double)(td[j] * td[j]);
        }
        if (g_exact_sums) {
            for (int e = 0; e < 36; ++e) H[e] = (float)dH[e];
            for (int e = 0; e < 6; ++e) g[e] = (float)dg[e];
            mse_sum = (float)dmse;
        }
        if (n_corr < 6) { rc = TCO_ALGORITHM; break; }                        /* :253-257 */
        float mse = mse_sum / (float)n_corr;
        float x[6];
        if (!tco_cholesky6_solve(H, g, x)) { if (!tco_lu6_solve(H, g, x)) { rc = TCO_ALGORITHM; break; } }
        float hx = x[0] / 2.0f, hy = x[1] / 2.0f, hz = x[2] / 2.0f;
        float qx[4] = { 1.0f * sinf(hx), 0.0f * sinf(hx), 0.0f * sinf(hx), cosf(hx) };
        float qy[4] = { 0.0f * sinf(hy), 1.0f * sinf(hy), 0.0f * sinf(hy), cosf(hy) };
        float qz[4] = { 0.0f * sinf(hz), 0.0f * sinf(hz), 1.0f * sinf(hz), cosf(hz) };
        float zy[4], rot[4], delta[7], nxt[7];
        quat_mul(qz, qy, zy); quat_mul(zy, qx, rot);
        delta[0] = rot[0]; delta[1] = rot[1]; delta[2] = rot[2]; delta[3] = rot[3]; delta[4] = x[3]; delta[5] = x[4]; delta[6] = x[5];
        tco_isometry_mul(delta, cur, nxt); memcpy(cur, nxt, sizeof(cur));
        if (fabsf(prev_mse - mse) < conv_thr) {                               /* :284-292 */
            memcpy(res->transform, cur, sizeof(cur)); res->mse = mse; res->iterations = it + 1; res->converged = 1;
            result_store_corr(res, pcs, pct, n_corr);
            done = 1; break;
        }
        prev_mse = mse;
        uint64_t *t1 = fcs; fcs = pcs; pcs = t1; t1 = fct; fct = pct; pct = t1; nfinal = n_corr;
    }
    if (rc == TCO_OK && !done) {
        memcpy(res->transform, cur, sizeof(cur)); res->mse = prev_mse; res->iterations = max_iters; res->converged = 0;
        result_store_corr(res, fcs, fct, nfinal);
    }
    free(cs); free(ct); free(ts); free(ti); free(td); free(pcs); free(pct); free(fcs); free(fct);
    tco_kdtree_free(tree);
    return rc;
}
