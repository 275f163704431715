// Combined issue limit at 16 single-wave workgroups per CU (dev tool): groups of NS scalar and NV vector
// instructions, interleaved, each chain dependent inside its own pipe only.  Does a CU reach
// 0.96 SALU + 1.0 VALU per cycle at once, or is there a lower combined ceiling?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define S1 "s_add_u32 %0, %0, 7\n"
#define S2 "s_lshr_b32 %1, %0, 3\n"
#define S3 "s_sub_u32 %0, %0, %1\n"
#define S4 "s_cselect_b32 %1, %1, %0\n"
#define V1 "v_add_u32 %2, %2, 3\n"
#define V2 "v_lshrrev_b32 %3, 3, %2\n"
#define V3 "v_sub_u32 %2, %2, %3\n"
#define V4 "v_min_u32 %3, %3, %2\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t s = seed, t = seed + 5, v = threadIdx.x + seed, w = v + 1;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(REP16(S1 S2 S3 S4 S1 S2 S3 S4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc");                         // 8 S
        if (MODE == 1) asm volatile(REP16(S1 S2 V1 S3 S4 V2 S1 S2 V3 S3 S4 V4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc");             // 8 S 4 V
        if (MODE == 2) asm volatile(REP16(S1 V1 S2 V2 S3 V3 S4 V4 S1 V1 S2 V2 S3 V3 S4 V4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc"); // 8 S 8 V
        if (MODE == 3) asm volatile(REP16(S1 V1 V2 S2 V3 S3 V4 V1 S4 V2 S1 V3 S2 V4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc");       // 6 S 8 V
        if (MODE == 4) asm volatile(REP16(S1 V1 V2 S2 V3 V4 S3 V1 V2 S4 V3 V4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc");             // 4 S 8 V
        if (MODE == 5) asm volatile(REP16(V1 V2 V3 V4 V1 V2 V3 V4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc");                         // 8 V
        // cross-pipe dependences as in a tree level: VALU compare -> VCC -> SALU -> SGPR operand of the next VALU
        if (MODE == 6) asm volatile(REP16("v_subrev_co_u32 %3, vcc, %0, %2\n s_cmp_lg_u32 vcc_lo, 0\n v_min_u32 %2, %2, %3\n s_cselect_b32 %0, %0, %1\n s_addc_u32 %1, %1, %1\n s_lshr_b32 s80, %0, 31\n")
                                    : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80");                                          // 4 S 2 V
    }
    out[blockIdx.x * 64 + threadIdx.x] = s + v + t + w;
}
template <int MODE>
double run(int per_cu, int iters, uint32_t *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), 4096, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), 4096, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 20000;
    const char *names[] = {"8 S", "8 S + 4 V", "8 S + 8 V", "6 S + 8 V", "4 S + 8 V", "8 V", "tree-level core 4 S + 2 V (cross-pipe)"};
    int ns[] = {8, 8, 8, 6, 4, 0, 4}, nv[] = {0, 4, 8, 8, 8, 8, 2};
    for (int mode = 0; mode < 7; mode++) {
        for (int pc : {8, 12, 16, 20}) {
            double ms = 0;
            switch (mode) {
            case 0: ms = run<0>(pc, iters, d); break; case 1: ms = run<1>(pc, iters, d); break;
            case 2: ms = run<2>(pc, iters, d); break; case 3: ms = run<3>(pc, iters, d); break;
            case 4: ms = run<4>(pc, iters, d); break; case 5: ms = run<5>(pc, iters, d); break;
            case 6: ms = run<6>(pc, iters, d); break; }
            double groups = (double)iters * 16 * pc;
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-40s waves/CU %2d: %7.2f ms  S/cycle/CU %.3f  V/cycle/CU %.3f  total %.3f\n", names[mode], pc, ms,
                   groups * ns[mode] / cyc, groups * nv[mode] / cyc, groups * (ns[mode] + nv[mode]) / cyc);
        }
    }
    return 0;
}
