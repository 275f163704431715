// Scalar-port model at 16 single-wave workgroups per CU (dev tool): do not-taken branches,
// v_readlane / v_writelane and plain VALU ops take issue slots away from SALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t s = seed, t = seed + 5, v = threadIdx.x + seed, w = v + 1, idx = 3;
    for (int i = 0; i < iters; i++) {
        // 8 dependent SALU per group
        if (MODE == 0) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n") : "+s"(s), "+s"(t)::"scc");
        // same + 1 never-taken branch
        if (MODE == 1) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n s_cmp_eq_u32 %0, 0x12345\n s_cbranch_scc1 1f\n1:\n") : "+s"(s), "+s"(t)::"scc");
        // same as 0 + s_cmp only (9 SALU)
        if (MODE == 2) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n s_cmp_eq_u32 %0, 0x12345\n") : "+s"(s), "+s"(t)::"scc");
        // 8 SALU + v_readlane + v_writelane
        if (MODE == 3) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n v_writelane_b32 %2, %0, 5\n v_readlane_b32 %1, %3, %4\n") : "+s"(s), "+s"(t), "+v"(v) : "v"(w), "s"(idx) : "scc");
        // 8 SALU + 2 plain VALU
        if (MODE == 4) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n v_add_u32 %2, %2, 3\n v_add_u32 %2, %2, 5\n") : "+s"(s), "+s"(t), "+v"(v)::"scc");
        // 8 SALU + 6 plain VALU
        if (MODE == 5) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n v_add_u32 %2, %2, 3\n s_sub_u32 %1, %0, %1\n v_add_u32 %2, %2, 3\n s_add_u32 %0, %0, %1\n v_add_u32 %2, %2, 3\n s_cselect_b32 %0, %0, %1\n v_add_u32 %2, %2, 3\n s_cselect_b32 %1, %1, %0\n v_add_u32 %2, %2, 3\n s_addc_u32 %0, %0, %0\n v_add_u32 %2, %2, 5\n") : "+s"(s), "+s"(t), "+v"(v)::"scc");
        // the real tree level: writelane, 6 core, addc, nchk, readlane
        if (MODE == 6) asm volatile(REP16("v_writelane_b32 %2, %1, 5\n s_lshr_b32 s80, %0, 11\n s_mul_i32 s80, s80, %1\n s_sub_u32 s81, %0, s80\n s_sub_u32 s82, %0, s80\n s_cselect_b32 %0, s80, s81\n s_cselect_b32 %0, %0, s82\n s_addc_u32 %4, %4, %4\n s_lshr_b32 s80, %0, 31\n s_cbranch_scc1 1f\n1:\n s_and_b32 %4, %4, 63\n v_readlane_b32 %1, %3, %4\n") : "+s"(s), "+s"(t), "+v"(v) : "v"(w), "s"(idx) : "scc", "s80", "s81", "s82");
        // candidate: bound for all 64 slots on the VALU, the lane select picks the bound (no s_mul, no record)
        if (MODE == 7) asm volatile(REP16("s_lshr_b32 s80, %0, 11\n v_mul_u32_u24 %2, s80, %3\n v_readlane_b32 s80, %2, %4\n s_sub_u32 s81, %0, s80\n s_sub_u32 s82, %0, s80\n s_cselect_b32 %0, s80, s81\n s_cselect_b32 %0, %0, s82\n s_addc_u32 %4, %4, %4\n s_lshr_b32 s80, %0, 31\n s_cbranch_scc1 1f\n1:\n s_and_b32 %4, %4, 63\n") : "+s"(s), "+s"(t), "+v"(v) : "v"(w), "s"(idx) : "scc", "s80", "s81", "s82");
        if (MODE == 8) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_nop 0\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n") : "+s"(s), "+s"(t)::"scc");
        if (MODE == 9) asm volatile(REP16("s_add_u32 %0, %0, 7\n s_lshr_b32 %1, %0, 3\n s_mul_i32 %0, %1, %0\n s_sub_u32 %1, %0, %1\n s_nop 1\n s_add_u32 %0, %0, %1\n s_cselect_b32 %0, %0, %1\n s_cselect_b32 %1, %1, %0\n s_addc_u32 %0, %0, %0\n") : "+s"(s), "+s"(t)::"scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = s + v + t;
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
    const char *names[] = {"8 SALU", "8 SALU + cmp + br-nt", "9 SALU", "8 SALU + wrlane + rdlane", "8 SALU + 2 VALU", "8 SALU + 6 VALU", "tree level (9S+br+2lane)", "tree level VALU bound", "8 SALU + s_nop 0", "8 SALU + s_nop 1"};
    int salu[] = {8, 9, 9, 8, 8, 8, 9, 8, 8, 8};
    for (int mode : {0, 6}) {
        for (int pc : {12, 16, 17, 18, 20, 24, 28, 32}) {
            double ms = 0;
            switch (mode) {
            case 0: ms = run<0>(pc, iters, d); break; case 1: ms = run<1>(pc, iters, d); break;
            case 2: ms = run<2>(pc, iters, d); break; case 3: ms = run<3>(pc, iters, d); break;
            case 4: ms = run<4>(pc, iters, d); break; case 5: ms = run<5>(pc, iters, d); break;
            case 6: ms = run<6>(pc, iters, d); break; case 7: ms = run<7>(pc, iters, d); break; case 8: ms = run<8>(pc, iters, d); break; case 9: ms = run<9>(pc, iters, d); break; }
            double groups = (double)iters * 16;
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-28s per_cu %2d: %.2f ms  cycles/group/wave %.1f  SALU/cycle/CU %.3f  groups/cycle/CU %.4f\n", names[mode], pc, ms,
                   cyc / groups, groups * salu[mode] * pc / cyc, groups * pc / cyc);
        }
    }
    return 0;
}
