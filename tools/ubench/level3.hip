// Cost of the block-gather tree level, single wave, by variant (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define WCORE "s_lshr_b32 s80, %[range], 11\n s_mul_i32 s80, s80, s86\n s_sub_u32 s81, %[range], s80\n s_sub_u32 s87, %[code], s80\n" \
              "s_cselect_b32 %[range], s80, s81\n s_cselect_b32 %[code], %[code], s87\n"
#define TAIL "s_lshr_b32 s80, %[range], 24\n s_cbranch_scc0 1f\n1:\n s_or_b32 %[range], %[range], 0x80000000\n s_and_b32 s88, s88, 31\n s_or_b32 s88, s88, 1\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t range = 0xFFFFFFFF, code = seed * 2654435761u;
    uint32_t v50 = 1024 + (threadIdx.x & 7);
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) // full new level
            asm volatile("s_mov_b32 s86, 1024\n s_mov_b32 s88, 1\n .rept 8\n"
                "v_writelane_b32 v54, s86, 3\n s_lshl1_add_u32 s85, s88, 1\n" WCORE "s_subb_u32 s88, s85, 0\n" TAIL
                "v_readlane_b32 s86, %[v50], s88\n .endr\n"
                : [range] "+s"(range), [code] "+s"(code) : [v50] "v"(v50) : "scc", "s80", "s81", "s85", "s86", "s87", "s88", "v54");
        if (MODE == 1) // no writelane
            asm volatile("s_mov_b32 s86, 1024\n s_mov_b32 s88, 1\n .rept 8\n"
                "s_lshl1_add_u32 s85, s88, 1\n" WCORE "s_subb_u32 s88, s85, 0\n" TAIL
                "v_readlane_b32 s86, %[v50], s88\n .endr\n"
                : [range] "+s"(range), [code] "+s"(code) : [v50] "v"(v50) : "scc", "s80", "s81", "s85", "s86", "s87", "s88", "v54");
        if (MODE == 2) // no readlane (p constant), no writelane: pure SALU
            asm volatile("s_mov_b32 s86, 1024\n s_mov_b32 s88, 1\n .rept 8\n"
                "s_lshl1_add_u32 s85, s88, 1\n" WCORE "s_subb_u32 s88, s85, 0\n" TAIL
                "s_add_u32 s86, s86, 1\n s_and_b32 s86, s86, 2047\n .endr\n"
                : [range] "+s"(range), [code] "+s"(code) : [v50] "v"(v50) : "scc", "s80", "s81", "s85", "s86", "s87", "s88", "v54");
        if (MODE == 3) // readlane + 2 s_nop between readlane and use
            asm volatile("s_mov_b32 s86, 1024\n s_mov_b32 s88, 1\n .rept 8\n"
                "s_lshl1_add_u32 s85, s88, 1\n s_lshr_b32 s80, %[range], 11\n s_nop 0\n s_nop 0\n s_mul_i32 s80, s80, s86\n s_sub_u32 s81, %[range], s80\n s_sub_u32 s87, %[code], s80\n"
                "s_cselect_b32 %[range], s80, s81\n s_cselect_b32 %[code], %[code], s87\n s_subb_u32 s88, s85, 0\n" TAIL
                "v_readlane_b32 s86, %[v50], s88\n .endr\n"
                : [range] "+s"(range), [code] "+s"(code) : [v50] "v"(v50) : "scc", "s80", "s81", "s85", "s86", "s87", "s88", "v54");
    }
    out[blockIdx.x * 64 + threadIdx.x] = range + code;
}
template <int MODE> double run(int pc, int iters, uint32_t *d)
{
    const int LDSB = pc <= 10 ? 16120 : (pc <= 16 ? 10000 : 7800);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * pc), dim3(64), LDSB, 0, d, 10, 1u); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * pc), dim3(64), LDSB, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 20000;
    const char *names[] = {"new level", "no writelane", "pure SALU (no readlane)", "readlane + nops"};
    int ninstr[] = {18, 17, 18, 19};
    for (int mode = 0; mode < 1; mode++)
        for (int pc : {4, 8, 12, 16, 20}) {
            double ms = mode == 0 ? run<0>(pc, iters, d) : mode == 1 ? run<1>(pc, iters, d) : mode == 2 ? run<2>(pc, iters, d) : run<3>(pc, iters, d);
            double cyc = ms * 1e-3 * 2.4e9 / (iters * 8.0);
            printf("%-26s per_cu %2d: cycles/level/wave %.1f (%.1f per instr)  cycles/level/CU %.1f\n", names[mode], pc, cyc, cyc / ninstr[mode], cyc / pc);
        }
    return 0;
}
