// dev microbenchmark (round 6): what a per-lane gather of two consecutive 3-float records from LDS costs a CU, by layout.
// Every lane walks its own run of records (start random per lane, two records per step, like the normals kernel's flattened walk):
//   A  AoS 12-byte pitch, 2 x ds_read_b96 (4-byte aligned addresses)
//   B  AoS 16-byte pitch, 2 x ds_read_b128
//   C  SoA x[] y[] z[],   3 x ds_read2_b32 (record j and j + 1 of one coordinate per instruction)
//   D  AoS 12-byte pitch, 3 x ds_read2_b32 (offsets 0 / 3: the same coordinate of two consecutive records)
//   E  AoS 12-byte pitch, ds_read_b128 + ds_read_b64 (six consecutive dwords; 4-byte aligned)
// 256-thread blocks, W blocks per CU (W waves per SIMD), LDS-array cycles per step = ticks * W / steps ... reported as ticks per step per wave
// and ns per step per CU.   hipcc -O3 --offload-arch=gfx950 tools/dev/micro/lds_gather.hip -o /tmp/lds_gather && /tmp/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kRecords = 3072;       // 36 KB at 12 B, 48 KB at 16 B
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int SPREAD>
__global__ void __launch_bounds__(256) gather_kernel(float *out, unsigned long long *ticks, int steps, uint32_t seed) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < kRecords * 4; i += 256) lds[i] = (float)(i * 7 % 1001) * 1e-3f;
    __syncthreads();
    // SPREAD 0: every lane starts at a random record; 1: neighbouring lanes start ~1.5 records apart (one row, lock step); 2: eight rows of eight lanes
    uint32_t h = (threadIdx.x + 1u) * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t j = SPREAD == 0 ? h % (kRecords - 600) : SPREAD == 1 ? (lane * 3u) / 2u + (h & 1u) : (lane >> 3) * 300u + ((lane & 7u) * 3u) / 2u;
    float ax = 0, ay = 0, az = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        if constexpr (KIND == 0) {
            f3 a, b; const uint32_t ad = j * 12u;
            asm volatile("ds_read_b96 %0, %2\n\tds_read_b96 %1, %2 offset:12\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(ad));
            ax += a.x + b.x; ay += a.y + b.y; az += a.z + b.z;
        } else if constexpr (KIND == 1) {
            f4 a, b; const uint32_t ad = j * 16u;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(ad));
            ax += a.x + b.x; ay += a.y + b.y; az += a.z + b.z;
        } else if constexpr (KIND == 2) {
            f2 x, y, z; const uint32_t ad = j * 4u;
            asm volatile("ds_read2_b32 %0, %3 offset0:0 offset1:1\n\tds_read2_b32 %1, %4 offset0:0 offset1:1\n\tds_read2_b32 %2, %5 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(ad), "v"(ad + kRecords * 4u), "v"(ad + kRecords * 8u));
            ax += x.x + x.y; ay += y.x + y.y; az += z.x + z.y;
        } else if constexpr (KIND == 3) {
            f2 x, y, z; const uint32_t ad = j * 12u;
            asm volatile("ds_read2_b32 %0, %3 offset0:0 offset1:3\n\tds_read2_b32 %1, %3 offset0:1 offset1:4\n\tds_read2_b32 %2, %3 offset0:2 offset1:5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(ad));
            ax += x.x + x.y; ay += y.x + y.y; az += z.x + z.y;
        } else {
            f4 a; f2 b; const uint32_t ad = j * 12u;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(ad));
            ax += a.x + a.w; ay += a.y + b.x; az += a.z + b.y;
        }
        j += 2u;
        j = j >= (uint32_t)(kRecords - 8) ? j - (uint32_t)(kRecords - 600) : j;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = ax + ay + az;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int SPREAD>
static int run(const char *name, int ncu) {
    const int steps = 20000;
    float *out; unsigned long long *ticks;
    CK(hipMalloc(&out, sizeof(float) * 256 * ncu * 4));
    CK(hipMalloc(&ticks, sizeof(unsigned long long) * 4 * ncu * 4));
    CK(hipFuncSetAttribute((const void *)gather_kernel<KIND, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%-44s", name);
    for (int W : {1, 2, 3}) {
        const size_t lds = (size_t)(160 * 1024 / W) & ~(size_t)255;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((gather_kernel<KIND, SPREAD>), dim3(ncu * W), dim3(256), lds, 0, out, ticks, 100, 777u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather_kernel<KIND, SPREAD>), dim3(ncu * W), dim3(256), lds, 0, out, ticks, steps, 777u);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> t(4 * (size_t)ncu * W);
        CK(hipMemcpy(t.data(), ticks, t.size() * sizeof(t[0]), hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end());
        // per CU: 4 W waves each take `steps` steps: ns of CU time per wave-step
        printf("  W=%d: %6.1f ticks/step/wave  %6.2f ns per wave-step per CU", W, (double)t[t.size() / 2] / steps, 1e6 * ms / ((double)steps * 4 * W));
    }
    printf("\n");
    CK(hipFree(out)); CK(hipFree(ticks));
    return 0;
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("two consecutive 3-float records per lane per step from LDS; W = waves per SIMD (256-thread blocks, W per CU)\n");
    if (run<0, 0>("A 12-B pitch 2 x ds_read_b96, random starts", ncu)) return 1;
    if (run<1, 0>("B 16-B pitch 2 x ds_read_b128, random starts", ncu)) return 1;
    if (run<2, 0>("C SoA 3 x ds_read2_b32, random starts", ncu)) return 1;
    if (run<3, 0>("D 12-B pitch 3 x ds_read2_b32, random starts", ncu)) return 1;
    if (run<4, 0>("E 12-B pitch b128 + b64, random starts", ncu)) return 1;
    if (run<0, 1>("A 12-B pitch 2 x ds_read_b96, one row", ncu)) return 1;
    if (run<1, 1>("B 16-B pitch 2 x ds_read_b128, one row", ncu)) return 1;
    if (run<2, 1>("C SoA 3 x ds_read2_b32, one row", ncu)) return 1;
    if (run<3, 1>("D 12-B pitch 3 x ds_read2_b32, one row", ncu)) return 1;
    if (run<4, 1>("E 12-B pitch b128 + b64, one row", ncu)) return 1;
    if (run<0, 2>("A 12-B pitch 2 x ds_read_b96, 8 rows of 8", ncu)) return 1;
    if (run<1, 2>("B 16-B pitch 2 x ds_read_b128, 8 rows of 8", ncu)) return 1;
    if (run<2, 2>("C SoA 3 x ds_read2_b32, 8 rows of 8", ncu)) return 1;
    if (run<3, 2>("D 12-B pitch 3 x ds_read2_b32, 8 rows of 8", ncu)) return 1;
    if (run<4, 2>("E 12-B pitch b128 + b64, 8 rows of 8", ncu)) return 1;
    return 0;
}
