// dev microbenchmark: what ONE wave64 VALU instruction costs a SIMD on gfx950, by instruction and by the
// number of waves that share the SIMD.  A dependence-free stream (16 independent registers per lane, 64 instructions per loop
// pass) runs in every wave of a grid that puts exactly W waves on every SIMD (blocks of 256 threads = one wave per SIMD, W blocks
// per CU, held to W by their dynamic LDS); cycles per wave-instruction on a SIMD = elapsed shader cycles / (W * instructions per wave).
//   hipcc -O3 --offload-arch=gfx950 tools/dev/micro/valu_price.hip -o /tmp/valu_price && /tmp/valu_price
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
template <int KIND>
__global__ void __launch_bounds__(256) stream_kernel(uint32_t *out, unsigned long long *ticks, int passes, uint32_t seed) {
    extern __shared__ uint32_t hold[];
    uint32_t a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed * (threadIdx.x + 1u) + (uint32_t)i * 0x9E3779B9u;
    uint32_t b = seed ^ 0x3f800000u, c = seed + threadIdx.x;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk[8], pkb = {1.0001f, 0.9999f}, pkc = {1e-3f, 2e-3f};
    for (int i = 0; i < 8; ++i) pk[i] = f2{(float)threadIdx.x + i, (float)seed};
    double dd[8], db = 1.0000001, dc = 1e-9;
    for (int i = 0; i < 8; ++i) dd[i] = (double)threadIdx.x + i;
    const unsigned long long msk = 0x5555555555555555ull * (seed | 1u);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p) {
#define FMA(i)  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MED(i)  asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define BFI(i)  asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ADD(i)  asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MUL(i)  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define CND(i)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
#define LSA(i)  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
#define CMP(i)  asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ANDB(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define XORB(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
#define MOVB(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MINF(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MEDF(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MEDI(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MAX3F(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MIN3U(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SUBF(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define CMPF(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
#define CNDS(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
#define AND3(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[(i) & 7]) : "v"(pkb), "v"(pkc));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[(i) & 7]) : "v"(pkb));
#define OPX40(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX41(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX42(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX43(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX44(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
#define OPX45(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
#define OPX46(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX47(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX48(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX49(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX50(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[i]));
#define OPX51(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX52(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX53(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX54(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX55(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX56(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(b));
#define OPX57(i) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));
#define OPX58(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX59(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OPX60(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[(i) & 7]) : "v"(db), "v"(dc));
#define OPX61(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[(i) & 7]) : "v"(db));
#define OPX62(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[(i) & 7]) : "v"(db));
#define OPX63(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
#define OPX64(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
#define OPX65(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define OPX66(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define OPX67(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OPX68(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(dd[(i) & 7]) : "v"(db));
#define PKF(i)  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(uint64_t *)&a[(i) & 14]) : "v"(*(uint64_t *)&a[(i) & 14]), "v"(*(uint64_t *)&a[(i) & 14]));
        if constexpr (KIND == 0) { REP16(FMA) REP16(FMA) REP16(FMA) REP16(FMA) }
        if constexpr (KIND == 1) { REP16(MED) REP16(MED) REP16(MED) REP16(MED) }
        if constexpr (KIND == 2) { REP16(BFI) REP16(BFI) REP16(BFI) REP16(BFI) }
        if constexpr (KIND == 3) { REP16(ADD) REP16(ADD) REP16(ADD) REP16(ADD) }
        if constexpr (KIND == 4) { REP16(MINU) REP16(MINU) REP16(MINU) REP16(MINU) }
        if constexpr (KIND == 5) { REP16(CND) REP16(CND) REP16(CND) REP16(CND) }
        if constexpr (KIND == 6) { REP16(LSA) REP16(LSA) REP16(LSA) REP16(LSA) }
        if constexpr (KIND == 7) { REP16(CMP) REP16(CMP) REP16(CMP) REP16(CMP) }
        if constexpr (KIND == 8) { REP16(MUL) REP16(MUL) REP16(MUL) REP16(MUL) }
#define Q4(OP) REP16(OP) REP16(OP) REP16(OP) REP16(OP)
        if constexpr (KIND == 10) { Q4(ADDU) }
        if constexpr (KIND == 11) { Q4(ANDB) }
        if constexpr (KIND == 12) { Q4(XORB) }
        if constexpr (KIND == 13) { Q4(LSHL) }
        if constexpr (KIND == 14) { Q4(MOVB) }
        if constexpr (KIND == 15) { Q4(MAXF) }
        if constexpr (KIND == 16) { Q4(MINF) }
        if constexpr (KIND == 17) { Q4(MEDF) }
        if constexpr (KIND == 18) { Q4(MEDI) }
        if constexpr (KIND == 19) { Q4(MAX3F) }
        if constexpr (KIND == 20) { Q4(MIN3U) }
        if constexpr (KIND == 21) { Q4(SUBF) }
        if constexpr (KIND == 22) { Q4(CMPF) }
        if constexpr (KIND == 23) { Q4(CNDS) }
        if constexpr (KIND == 24) { Q4(AND3) }
        if constexpr (KIND == 25) { Q4(FMAC) }
        if constexpr (KIND == 26) { Q4(PKFMA) }
        if constexpr (KIND == 27) { Q4(PKADD) }
        if constexpr (KIND == 40) { Q4(OPX40) }
        if constexpr (KIND == 41) { Q4(OPX41) }
        if constexpr (KIND == 42) { Q4(OPX42) }
        if constexpr (KIND == 43) { Q4(OPX43) }
        if constexpr (KIND == 44) { Q4(OPX44) }
        if constexpr (KIND == 45) { Q4(OPX45) }
        if constexpr (KIND == 46) { Q4(OPX46) }
        if constexpr (KIND == 47) { Q4(OPX47) }
        if constexpr (KIND == 48) { Q4(OPX48) }
        if constexpr (KIND == 49) { Q4(OPX49) }
        if constexpr (KIND == 50) { Q4(OPX50) }
        if constexpr (KIND == 51) { Q4(OPX51) }
        if constexpr (KIND == 52) { Q4(OPX52) }
        if constexpr (KIND == 53) { Q4(OPX53) }
        if constexpr (KIND == 54) { Q4(OPX54) }
        if constexpr (KIND == 55) { Q4(OPX55) }
        if constexpr (KIND == 56) { Q4(OPX56) }
        if constexpr (KIND == 57) { Q4(OPX57) }
        if constexpr (KIND == 58) { Q4(OPX58) }
        if constexpr (KIND == 59) { Q4(OPX59) }
        if constexpr (KIND == 60) { Q4(OPX60) }
        if constexpr (KIND == 61) { Q4(OPX61) }
        if constexpr (KIND == 62) { Q4(OPX62) }
        if constexpr (KIND == 63) { Q4(OPX63) }
        if constexpr (KIND == 64) { Q4(OPX64) }
        if constexpr (KIND == 65) { Q4(OPX65) }
        if constexpr (KIND == 66) { Q4(OPX66) }
        if constexpr (KIND == 67) { Q4(OPX67) }
        if constexpr (KIND == 68) { Q4(OPX68) }
        // the same mix with the list as f32 keys (v_med3_f32) and float compares
        if constexpr (KIND == 28) { REP16(MEDF) REP16(MEDF) MEDF(0) MEDF(1) MEDF(2) MEDF(3) ADD(4) MUL(5) ADD(6) MUL(7) ADD(8) MUL(9) ADD(10) MUL(11) ADD(12) MUL(13) ADD(14) MUL(15)
                                   BFI(0) BFI(1) BFI(2) BFI(3) CMP(4) CNDS(5) CMP(6) CNDS(7) CMP(8) CNDS(9) CMP(10) CNDS(11) LSA(12) LSA(13) LSA(14) LSA(15) }
        // the normals walk's mix: 36 med3 + 12 f32 (sub/mul/add) + 4 bfi + 8 cmp/cndmask + 4 shifts/adds = 64
        if constexpr (KIND == 9) { REP16(MED) REP16(MED) MED(0) MED(1) MED(2) MED(3) ADD(4) MUL(5) ADD(6) MUL(7) ADD(8) MUL(9) ADD(10) MUL(11) ADD(12) MUL(13) ADD(14) MUL(15)
                                   BFI(0) BFI(1) BFI(2) BFI(3) CMP(4) CND(5) CMP(6) CND(7) CMP(8) CND(9) CMP(10) CND(11) LSA(12) LSA(13) LSA(14) LSA(15) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) s ^= a[i];
    for (int i = 0; i < 8; ++i) s ^= (uint32_t)__double_as_longlong(dd[i]);
    for (int i = 0; i < 8; ++i) s ^= __float_as_uint(pk[i].x) ^ __float_as_uint(pk[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = s + hold[threadIdx.x & 1];
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static int run(const char *name, int ncu) {
    const int passes = 4000;                    // x 64 instructions per wave
    uint32_t *out; unsigned long long *ticks;
    CK(hipMalloc(&out, sizeof(uint32_t) * 256 * ncu * 8));
    CK(hipMalloc(&ticks, sizeof(unsigned long long) * 4 * ncu * 8));
    CK(hipFuncSetAttribute((const void *)stream_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%-14s", name);
    for (int W : {1, 2, 4, 8}) {
        const size_t lds = (size_t)(160 * 1024 / W) & ~(size_t)255;         // W blocks fit a CU, W + 1 do not
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(stream_kernel<KIND>, dim3(ncu * W), dim3(256), lds, 0, out, ticks, 50, 12345u);      // warm-up
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel<KIND>, dim3(ncu * W), dim3(256), lds, 0, out, ticks, passes, 12345u);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> t(4 * (size_t)ncu * W);
        CK(hipMemcpy(t.data(), ticks, t.size() * sizeof(t[0]), hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end());
        const double med = (double)t[t.size() / 2], ninst = (double)passes * 64.0;
        // s_memtime ticks per wave-instruction per SIMD; wall ns per wave-instruction per SIMD
        printf("  W=%d: %5.2f tick  %5.3f ns", W, med / (W * ninst), 1e6 * ms / (W * ninst));
    }
    printf("\n");
    CK(hipFree(out)); CK(hipFree(ticks));
    return 0;
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("%s  CUs %d  clock %d kHz  (tick = s_memtime; ns = event time / (W * instructions per wave): at 2.4 GHz one cycle = 0.417 ns)\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    const int ncu = pr.multiProcessorCount;
    if (run<0>("v_fma_f32", ncu)) return 1;
    if (run<1>("v_med3_u32", ncu)) return 1;
    if (run<2>("v_bfi_b32", ncu)) return 1;
    if (run<3>("v_add_f32", ncu)) return 1;
    if (run<8>("v_mul_f32", ncu)) return 1;
    if (run<4>("v_min_u32", ncu)) return 1;
    if (run<5>("v_cndmask_b32", ncu)) return 1;
    if (run<6>("v_lshl_add_u32", ncu)) return 1;
    if (run<7>("v_cmp_lt_u32", ncu)) return 1;
    if (run<9>("walk mix", ncu)) return 1;
    if (run<28>("walk mix f32", ncu)) return 1;
    if (run<10>("v_add_u32", ncu)) return 1;
    if (run<11>("v_and_b32", ncu)) return 1;
    if (run<12>("v_xor_b32", ncu)) return 1;
    if (run<13>("v_lshlrev_b32", ncu)) return 1;
    if (run<14>("v_mov_b32", ncu)) return 1;
    if (run<15>("v_max_f32", ncu)) return 1;
    if (run<16>("v_min_f32", ncu)) return 1;
    if (run<17>("v_med3_f32", ncu)) return 1;
    if (run<18>("v_med3_i32", ncu)) return 1;
    if (run<19>("v_max3_f32", ncu)) return 1;
    if (run<20>("v_min3_u32", ncu)) return 1;
    if (run<21>("v_sub_f32", ncu)) return 1;
    if (run<22>("v_cmp_lt_f32", ncu)) return 1;
    if (run<23>("v_cndmask sgpr", ncu)) return 1;
    if (run<24>("v_and_or_b32", ncu)) return 1;
    if (run<25>("v_fmac_f32", ncu)) return 1;
    if (run<26>("v_pk_fma_f32", ncu)) return 1;
    if (run<27>("v_pk_add_f32", ncu)) return 1;
    if (run<40>("v_sub_u32", ncu)) return 1;
    if (run<41>("v_or_b32", ncu)) return 1;
    if (run<42>("v_mul_u32_u24", ncu)) return 1;
    if (run<43>("v_mul_lo_u32", ncu)) return 1;
    if (run<44>("v_ashrrev_i32", ncu)) return 1;
    if (run<45>("v_lshrrev_b32", ncu)) return 1;
    if (run<46>("v_max_u32", ncu)) return 1;
    if (run<47>("v_min_i32", ncu)) return 1;
    if (run<48>("v_mul_hi_u32", ncu)) return 1;
    if (run<49>("v_mad_u32_u24", ncu)) return 1;
    if (run<50>("v_bfe_u32", ncu)) return 1;
    if (run<51>("v_perm_b32", ncu)) return 1;
    if (run<52>("v_alignbit_b32", ncu)) return 1;
    if (run<53>("v_add3_u32", ncu)) return 1;
    if (run<54>("v_or3_b32", ncu)) return 1;
    if (run<55>("v_xad_u32", ncu)) return 1;
    if (run<56>("v_add_lshl_u32", ncu)) return 1;
    if (run<57>("v_lshl_or_b32", ncu)) return 1;
    if (run<58>("v_sad_u32", ncu)) return 1;
    if (run<59>("v_mad_i32_i24", ncu)) return 1;
    if (run<60>("v_fma_f64", ncu)) return 1;
    if (run<61>("v_add_f64", ncu)) return 1;
    if (run<62>("v_mul_f64", ncu)) return 1;
    if (run<63>("v_cvt_f32_u32", ncu)) return 1;
    if (run<64>("v_cvt_u32_f32", ncu)) return 1;
    if (run<65>("v_rcp_f32", ncu)) return 1;
    if (run<66>("v_sqrt_f32", ncu)) return 1;
    if (run<67>("v_mul_f32_e64", ncu)) return 1;
    if (run<68>("v_max_f64", ncu)) return 1;
    return 0;
}
