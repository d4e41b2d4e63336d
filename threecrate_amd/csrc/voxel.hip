// voxel.hip -- voxel_grid_filter (threecrate-algorithms/src/filtering.rs:38-133): one centroid per
// occupied voxel, f64 accumulation, keys = floor((p - bbox_min) / voxel_size) per axis.
//
// The reference folds the cloud into a HashMap in input order, so every voxel's f64 sum runs in
// ascending original index and the output order is unspecified.  Here: dense voxel grid (x-major
// linear id => output sorted by (kx, ky, kz)), counting sort by voxel id with the arrival-rank
// scatter + stable re-rank of grid.hip (points of a voxel end up in ascending original index), then
// one lane per occupied voxel folds its points in that order in f64: bit-identical centroids.
#include "tc_internal.h"

#include <algorithm>
#include <cmath>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace tc {

struct VoxGeom {
    float minx, miny, minz, voxel;
    int gx, gy, gz;
    uint32_t ncell;
};

__device__ __forceinline__ uint32_t voxel_id(const VoxGeom &v, float x, float y, float z) {
    // filtering.rs:96-101: ((p - min) / voxel_size).floor() as i32  (true division, not * 1/voxel)
    int kx = (int)floorf((x - v.minx) / v.voxel), ky = (int)floorf((y - v.miny) / v.voxel),
        kz = (int)floorf((z - v.minz) / v.voxel);
    kx = min(max(kx, 0), v.gx - 1); ky = min(max(ky, 0), v.gy - 1); kz = min(max(kz, 0), v.gz - 1);
    return ((uint32_t)kx * v.gy + ky) * v.gz + kz;       // x-major: ascending id = (kx, ky, kz) order
}

__global__ void __launch_bounds__(256) vox_hist_kernel(const float *__restrict__ xyz, uint32_t n, VoxGeom v,
                                                      uint32_t *__restrict__ cell_of, uint32_t *__restrict__ hist,
                                                      uint32_t *__restrict__ arrival) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = voxel_id(v, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
    cell_of[i] = c;
    arrival[i] = atomicAdd(&hist[c], 1u);
}

__global__ void __launch_bounds__(256) vox_scatter_kernel(const uint32_t *__restrict__ cell_of, uint32_t n,
                                                         const uint32_t *__restrict__ cell_start,
                                                         const uint32_t *__restrict__ arrival, uint32_t *__restrict__ slot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    slot[cell_start[cell_of[i]] + arrival[i]] = i;
}

// stable order inside the voxel: ascending original index
__global__ void __launch_bounds__(256) vox_rank_kernel(uint32_t n, const uint32_t *__restrict__ cell_of,
                                                      const uint32_t *__restrict__ cell_start,
                                                      const uint32_t *__restrict__ slot, uint32_t *__restrict__ order) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t i = slot[p];
    const uint32_t c = cell_of[i];
    const uint32_t s = cell_start[c], e = cell_start[c + 1];
    uint32_t rank = 0;
    for (uint32_t j = s; j < e; ++j) rank += (slot[j] < i) ? 1u : 0u;
    order[s + rank] = i;
}

__global__ void __launch_bounds__(256) vox_flag_kernel(const uint32_t *__restrict__ cell_start, uint32_t ncell,
                                                      uint32_t *__restrict__ flag) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    flag[c] = (cell_start[c + 1] > cell_start[c]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) vox_centroid_kernel(const float *__restrict__ xyz, const uint32_t *__restrict__ cell_start,
                                                          uint32_t ncell, const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ outpos, float *__restrict__ out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t s = cell_start[c], e = cell_start[c + 1];
    if (e == s) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;          // filtering.rs:108-118
    for (uint32_t j = s; j < e; ++j) {
        const size_t i = order[j];
        sx += (double)xyz[3 * i]; sy += (double)xyz[3 * i + 1]; sz += (double)xyz[3 * i + 2];
    }
    const double inv = 1.0 / (double)(e - s);      // filtering.rs:122-128
    float *o = out + 3 * (size_t)outpos[c];
    o[0] = (float)(sx * inv); o[1] = (float)(sy * inv); o[2] = (float)(sz * inv);
}

// ---- sort path: bounding boxes too large for a dense grid, or many points per voxel ---------------
// key = (kx, ky, kz) packed into bx + by + bz bits; a stable LSD radix sort of (key, original index) leaves the
// voxels in (kx, ky, kz) order and the points of a voxel in ascending original index, like the dense path.
struct VoxBits {
    float minx, miny, minz, voxel;
    int sy, sz;             // shifts: key = kx << sx | ky << sz... (kz in the low bits)
    int gx, gy, gz;
};

__global__ void __launch_bounds__(256) vox_key_kernel(const float *__restrict__ xyz, uint32_t n, VoxBits v,
                                                     uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // filtering.rs:96-101: ((p - min) / voxel_size).floor() as i32
    int kx = (int)floorf((xyz[3 * (size_t)i] - v.minx) / v.voxel), ky = (int)floorf((xyz[3 * (size_t)i + 1] - v.miny) / v.voxel),
        kz = (int)floorf((xyz[3 * (size_t)i + 2] - v.minz) / v.voxel);
    kx = min(max(kx, 0), v.gx - 1); ky = min(max(ky, 0), v.gy - 1); kz = min(max(kz, 0), v.gz - 1);
    keys[i] = ((uint64_t)kx << v.sy) | ((uint64_t)ky << v.sz) | (uint64_t)kz;
    idx[i] = i;
}

__global__ void __launch_bounds__(256) vox_head_kernel(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ head) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    head[p] = (p == 0 || keys[p] != keys[p - 1]) ? 1u : 0u;
}

// first sorted position of every voxel (vstart[M] = n)
__global__ void __launch_bounds__(256) vox_starts_kernel(uint32_t n, const uint32_t *__restrict__ head, const uint32_t *__restrict__ outpos,
                                                        uint32_t *__restrict__ vstart) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (head[p]) vstart[outpos[p]] = p;
    if (p == n - 1) vstart[outpos[n]] = n;
}

// one lane per voxel folds the voxel's points in sorted = original order (filtering.rs:108-118), eight gathers in
// flight at a time: the adds stay sequential, only the loads overlap
constexpr uint32_t kLongVoxel = 96;     // points; longer voxels go to the wave-per-voxel kernel

__global__ void __launch_bounds__(256) vox_centroid_sorted_kernel(const float *__restrict__ xyz, const uint32_t *__restrict__ order,
                                                                 const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ n_vox,
                                                                 float *__restrict__ out, uint32_t *__restrict__ long_list,
                                                                 uint32_t *__restrict__ long_count) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *n_vox) return;
    const uint32_t s = vstart[v], e = vstart[v + 1];
    if (e - s > kLongVoxel) { long_list[atomicAdd(long_count, 1u)] = v; return; }
    double sx = 0.0, sy = 0.0, sz = 0.0;
    uint32_t j = s;
    for (; j + 8 <= e; j += 8) {
        size_t i[8];
        float x[8], y[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) i[u] = order[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = xyz[3 * i[u]]; y[u] = xyz[3 * i[u] + 1]; z[u] = xyz[3 * i[u] + 2]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { sx += (double)x[u]; sy += (double)y[u]; sz += (double)z[u]; }
    }
    for (; j < e; ++j) {
        const size_t i = order[j];
        sx += (double)xyz[3 * i]; sy += (double)xyz[3 * i + 1]; sz += (double)xyz[3 * i + 2];
    }
    const double inv = 1.0 / (double)(e - s);      // filtering.rs:122-128
    float *o = out + 3 * (size_t)v;
    o[0] = (float)(sx * inv); o[1] = (float)(sy * inv); o[2] = (float)(sz * inv);
}

// Voxels with many points (0.2 m voxels on a depth frame: ~1000 each): one WAVE per voxel.  The 64 lanes gather 64
// consecutive points of the voxel at once; the f64 adds stay strictly sequential in sorted = original order (every lane
// runs the same chain on broadcast values), so the centroid has the bits of the reference's fold.
__global__ void __launch_bounds__(256) vox_centroid_long_kernel(const float *__restrict__ xyz, const uint32_t *__restrict__ order,
                                                               const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ long_list,
                                                               const uint32_t *__restrict__ long_count, float *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t total = *long_count;
    for (uint32_t entry = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); entry < total; entry += nwaves) {
        const uint32_t v = long_list[entry];
        const uint32_t s = vstart[v], e = vstart[v + 1];
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (uint32_t base = s; base < e; base += 64) {
            const uint32_t j = base + lane;
            float x = 0.0f, y = 0.0f, z = 0.0f;
            if (j < e) {
                const size_t i = order[j];
                x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2];
            }
            const int cnt = (int)min(64u, e - base);
            for (int l = 0; l < cnt; ++l) {        // l is wave-uniform: v_readlane with a scalar lane select
                sx += (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
                sy += (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), l));
                sz += (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), l));
            }
        }
        if (lane == 0) {
            const double inv = 1.0 / (double)(e - s);      // filtering.rs:122-128
            float *o = out + 3 * (size_t)v;
            o[0] = (float)(sx * inv); o[1] = (float)(sy * inv); o[2] = (float)(sz * inv);
        }
    }
}

static int bits_for(double dim) {
    int b = 1;
    while ((double)(1ull << b) < dim) ++b;
    return b;
}

static tc_status voxel_filter_sorted(tc_context *ctx, const float *d_xyz, size_t n, float voxel, const float mn[3], const double dims[3],
                                     float *d_out, size_t *n_out) {
    hipStream_t st = ctx->stream;
    for (int c = 0; c < 3; ++c)
        if (!(dims[c] < 2097152.0)) return fail(ctx, TC_UNSUPPORTED, "voxel_grid_filter: more than 2^21 voxels along one axis of the bounding box");
    VoxBits v;
    v.minx = mn[0]; v.miny = mn[1]; v.minz = mn[2]; v.voxel = voxel;
    v.gx = (int)dims[0]; v.gy = (int)dims[1]; v.gz = (int)dims[2];
    const int bx = bits_for(dims[0]), by = bits_for(dims[1]), bz = bits_for(dims[2]);
    v.sz = bz; v.sy = by + bz;
    DeviceIndex &ix = ctx->vox_index;
    const uint32_t n32 = (uint32_t)n;
    const int nb = (int)((n + 255) / 256);
    if (tc_status s = ensure(ctx, ix.cell_of, n * sizeof(uint64_t))) return s;            // keys
    if (tc_status s = ensure(ctx, ix.slot, n * sizeof(uint64_t))) return s;               // sorted keys
    if (tc_status s = ensure(ctx, ix.arrival, n * sizeof(uint32_t))) return s;            // indices
    if (tc_status s = ensure(ctx, ix.pts, n * sizeof(uint32_t))) return s;                // order[] = sorted indices
    if (tc_status s = ensure(ctx, ix.fill, n * sizeof(uint32_t))) return s;               // voxel heads
    if (tc_status s = ensure(ctx, ix.cell_start, (n + 2) * sizeof(uint32_t))) return s;   // output slot of a head, [n + 1] = long-voxel counter
    if (tc_status s = ensure(ctx, ctx->overflow, (n + 1) * sizeof(uint32_t))) return s;   // first sorted position of a voxel
    uint64_t *keys = (uint64_t *)ix.cell_of.p, *keys_sorted = (uint64_t *)ix.slot.p;
    uint32_t *idx = (uint32_t *)ix.arrival.p, *order = (uint32_t *)ix.pts.p, *head = (uint32_t *)ix.fill.p, *outpos = (uint32_t *)ix.cell_start.p;
    ProfScope ps(ctx, "voxel_grid_filter_sorted");
    hipLaunchKernelGGL(vox_key_kernel, dim3(nb), dim3(256), 0, st, d_xyz, n32, v, keys, idx);
    size_t temp_bytes = 0;
    TC_HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys_sorted, idx, order, n, 0u, (unsigned)(bx + by + bz), st));
    if (tc_status s = ensure(ctx, ix.normals, temp_bytes)) return s;
    TC_HIP_TRY(ctx, rocprim::radix_sort_pairs(ix.normals.p, temp_bytes, keys, keys_sorted, idx, order, n, 0u, (unsigned)(bx + by + bz), st));
    hipLaunchKernelGGL(vox_head_kernel, dim3(nb), dim3(256), 0, st, (const uint64_t *)keys_sorted, n32, head);
    if (tc_status s = exclusive_scan_u32(ctx, head, n32, outpos, ix.blocksum)) return s;
    uint32_t *vstart = (uint32_t *)ctx->overflow.p;
    hipLaunchKernelGGL(vox_starts_kernel, dim3(nb), dim3(256), 0, st, n32, (const uint32_t *)head, (const uint32_t *)outpos, vstart);
    TC_HIP_TRY(ctx, hipMemsetAsync(outpos + n + 1, 0, sizeof(uint32_t), st));
    uint32_t *long_list = head;          // the head flags are no longer needed
    hipLaunchKernelGGL(vox_centroid_sorted_kernel, dim3(nb), dim3(256), 0, st, d_xyz, (const uint32_t *)order, (const uint32_t *)vstart,
                       (const uint32_t *)(outpos + n), d_out, long_list, outpos + n + 1);
    hipLaunchKernelGGL(vox_centroid_long_kernel, dim3(1024), dim3(256), 0, st, d_xyz, (const uint32_t *)order, (const uint32_t *)vstart,
                       (const uint32_t *)long_list, (const uint32_t *)(outpos + n + 1), d_out);
    uint32_t *hcount = (uint32_t *)((char *)ctx->pinned + 1024);
    TC_HIP_TRY(ctx, hipMemcpyAsync(hcount, outpos + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    TC_HIP_TRY(ctx, hipGetLastError());
    *n_out = *hcount;
    return TC_OK;
}

tc_status voxel_filter_device(tc_context *ctx, const float *d_xyz, size_t n, float voxel, float *d_out, size_t *n_out) {
    hipStream_t st = ctx->stream;
    float mn[3], mx[3];
    if (tc_status s = cloud_bbox(ctx, d_xyz, n, mn, mx)) return s;
    for (int c = 0; c < 3; ++c)
        if (!(mn[c] <= mx[c]) || !std::isfinite(mn[c]) || !std::isfinite(mx[c]))
            return fail(ctx, TC_INVALID_DATA, "voxel_grid_filter: non-finite coordinates");
    VoxGeom v;
    v.minx = mn[0]; v.miny = mn[1]; v.minz = mn[2]; v.voxel = voxel;
    double dims[3];
    for (int c = 0; c < 3; ++c) dims[c] = std::floor((double)((mx[c] - mn[c]) / voxel)) + 1.0;
    const double ncd = dims[0] * dims[1] * dims[2];
    // the dense grid pays O(voxels in the box) and a quadratic re-rank inside a voxel: boxes with more than 2^25
    // voxels (0.05 m voxels on a LiDAR sweep) or many points per voxel (0.2 m voxels on a depth frame) are sorted
    if (!(ncd < 33554432.0) || (double)n > 8.0 * ncd) return voxel_filter_sorted(ctx, d_xyz, n, voxel, mn, dims, d_out, n_out);
    v.gx = (int)dims[0]; v.gy = (int)dims[1]; v.gz = (int)dims[2];
    v.ncell = (uint32_t)ncd;
    DeviceIndex &ix = ctx->vox_index;
    const uint32_t n32 = (uint32_t)n;
    const int nb = (int)((n + 255) / 256), ncb = (int)((v.ncell + 255) / 256);
    if (tc_status s = ensure(ctx, ix.cell_of, n * sizeof(uint32_t))) return s;
    if (tc_status s = ensure(ctx, ix.slot, n * sizeof(uint32_t))) return s;
    if (tc_status s = ensure(ctx, ix.arrival, n * sizeof(uint32_t))) return s;
    if (tc_status s = ensure(ctx, ix.pts, n * sizeof(uint32_t))) return s;                       // order[]
    if (tc_status s = ensure(ctx, ix.fill, (size_t)v.ncell * sizeof(uint32_t))) return s;        // histogram, then flags
    if (tc_status s = ensure(ctx, ix.cell_start, ((size_t)v.ncell + 1) * sizeof(uint32_t))) return s;
    if (tc_status s = ensure(ctx, ctx->overflow, ((size_t)v.ncell + 1) * sizeof(uint32_t))) return s;   // output slots
    TC_HIP_TRY(ctx, hipMemsetAsync(ix.fill.p, 0, (size_t)v.ncell * sizeof(uint32_t), st));
    ProfScope ps(ctx, "voxel_grid_filter");
    hipLaunchKernelGGL(vox_hist_kernel, dim3(nb), dim3(256), 0, st, d_xyz, n32, v, (uint32_t *)ix.cell_of.p, (uint32_t *)ix.fill.p,
                       (uint32_t *)ix.arrival.p);
    if (tc_status s = exclusive_scan_u32(ctx, (const uint32_t *)ix.fill.p, v.ncell, (uint32_t *)ix.cell_start.p, ix.blocksum)) return s;
    hipLaunchKernelGGL(vox_scatter_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)ix.cell_of.p, n32,
                       (const uint32_t *)ix.cell_start.p, (const uint32_t *)ix.arrival.p, (uint32_t *)ix.slot.p);
    hipLaunchKernelGGL(vox_rank_kernel, dim3(nb), dim3(256), 0, st, n32, (const uint32_t *)ix.cell_of.p,
                       (const uint32_t *)ix.cell_start.p, (const uint32_t *)ix.slot.p, (uint32_t *)ix.pts.p);
    hipLaunchKernelGGL(vox_flag_kernel, dim3(ncb), dim3(256), 0, st, (const uint32_t *)ix.cell_start.p, v.ncell, (uint32_t *)ix.fill.p);
    if (tc_status s = exclusive_scan_u32(ctx, (const uint32_t *)ix.fill.p, v.ncell, (uint32_t *)ctx->overflow.p, ix.blocksum)) return s;
    hipLaunchKernelGGL(vox_centroid_kernel, dim3(ncb), dim3(256), 0, st, d_xyz, (const uint32_t *)ix.cell_start.p, v.ncell,
                       (const uint32_t *)ix.pts.p, (const uint32_t *)ctx->overflow.p, d_out);
    uint32_t *hcount = (uint32_t *)((char *)ctx->pinned + 1024);
    TC_HIP_TRY(ctx, hipMemcpyAsync(hcount, (uint32_t *)ctx->overflow.p + v.ncell, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    TC_HIP_TRY(ctx, hipGetLastError());
    *n_out = *hcount;
    return TC_OK;
}

}  // namespace tc


// ---- range_filter (kiss_icp.rs:56-70): keep points with min_r^2 <= |p|^2 <= max_r^2, order preserved ----
namespace tc {

__global__ void __launch_bounds__(256) range_flag_kernel(const float *__restrict__ xyz, uint32_t n, float min_sq, float max_sq,
                                                        uint32_t *__restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    const float r2 = x * x + y * y + z * z;                 // magnitude_squared, no FMA (-ffp-contract=off)
    flag[i] = (r2 >= min_sq && r2 <= max_sq) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) range_compact_kernel(const float *__restrict__ xyz, uint32_t n, const uint32_t *__restrict__ flag,
                                                           const uint32_t *__restrict__ off, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint32_t o = off[i];
    out[3 * (size_t)o] = xyz[3 * (size_t)i]; out[3 * (size_t)o + 1] = xyz[3 * (size_t)i + 1]; out[3 * (size_t)o + 2] = xyz[3 * (size_t)i + 2];
}

tc_status range_filter_device(tc_context *ctx, const float *d_xyz, size_t n, float min_range, float max_range, float *d_out, size_t *n_out) {
    *n_out = 0;
    if (n == 0) return TC_OK;
    if (n >= 0xFFFFFFF0ull) return fail(ctx, TC_UNSUPPORTED, "more than 2^32 points");
    DeviceIndex &ix = ctx->vox_index;
    if (tc_status s = ensure(ctx, ix.fill, n * sizeof(uint32_t))) return s;
    if (tc_status s = ensure(ctx, ix.cell_start, (n + 1) * sizeof(uint32_t))) return s;
    hipStream_t st = ctx->stream;
    const unsigned nb = (unsigned)((n + 255) / 256);
    ProfScope ps(ctx, "range_filter");
    hipLaunchKernelGGL(range_flag_kernel, dim3(nb), dim3(256), 0, st, d_xyz, (uint32_t)n, min_range * min_range, max_range * max_range,
                       (uint32_t *)ix.fill.p);
    if (tc_status s = exclusive_scan_u32(ctx, (const uint32_t *)ix.fill.p, (uint32_t)n, (uint32_t *)ix.cell_start.p, ix.blocksum)) return s;
    hipLaunchKernelGGL(range_compact_kernel, dim3(nb), dim3(256), 0, st, d_xyz, (uint32_t)n, (const uint32_t *)ix.fill.p,
                       (const uint32_t *)ix.cell_start.p, d_out);
    uint32_t *hcount = (uint32_t *)((char *)ctx->pinned + 1024);
    TC_HIP_TRY(ctx, hipMemcpyAsync(hcount, (const uint32_t *)ix.cell_start.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    TC_HIP_TRY(ctx, hipStreamSynchronize(st));
    *n_out = *hcount;
    return TC_OK;
}

}   // namespace tc
