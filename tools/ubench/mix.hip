// Scalar-flavour vs vector-flavour tree level, and a mix of both on one CU (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// scalar flavour level (as in xlz_fastpath.inc)
#define SLEVEL \
    "v_writelane_b32 v54, s86, 3\n s_lshl1_add_u32 s85, s88, 1\n" \
    "s_lshr_b32 s80, %[range], 11\n s_mul_i32 s80, s80, s86\n s_sub_u32 s81, %[range], s80\n s_sub_u32 s87, %[code], s80\n" \
    "s_cselect_b32 %[range], s80, s81\n s_cselect_b32 %[code], %[code], s87\n s_subb_u32 s88, s85, 0\n" \
    "s_lshr_b32 s80, %[range], 24\n s_cbranch_scc0 1f\n1:\n" \
    "s_or_b32 %[range], %[range], 0x80000000\n s_and_b32 s88, s88, 31\n s_or_b32 s88, s88, 1\n" \
    "v_readlane_b32 s86, %[v50], s88\n"
// vector flavour level: range/code/M uniform in VGPRs
#define VLEVEL \
    "v_writelane_b32 v54, s86, 3\n v_lshl_add_u32 v46, %[vm], 1, 1\n" \
    "v_lshrrev_b32 v45, 11, %[vr]\n v_mul_u32_u24 v45, s86, v45\n v_sub_co_u32 v44, vcc, %[vc], v45\n v_sub_u32 v43, %[vr], v45\n" \
    "v_cndmask_b32 %[vr], v43, v45, vcc\n v_cndmask_b32 %[vc], v44, %[vc], vcc\n v_subb_co_u32 %[vm], vcc, v46, 0, vcc\n" \
    "v_cmp_gt_u32 vcc, s89, %[vr]\n s_cbranch_vccnz 1f\n1:\n" \
    "v_or_b32 %[vr], 0x80000000, %[vr]\n v_and_b32 %[vm], 31, %[vm]\n v_or_b32 %[vm], 1, %[vm]\n" \
    "v_readfirstlane_b32 s88, %[vm]\n s_nop 3\n v_readlane_b32 s86, %[v50], s88\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t range = 0xFFFFFFFF, code = seed * 2654435761u;
    uint32_t v50 = 1024 + (threadIdx.x & 7), vr = range, vc = code, vm = 1;
    const bool vflav = MODE == 1 || (MODE == 2 && (blockIdx.x & 1));
    if (!vflav) {
        for (int i = 0; i < iters; i++)
            asm volatile("s_mov_b32 s86, 1024\n s_mov_b32 s88, 1\n .rept 8\n" SLEVEL ".endr\n"
                : [range] "+s"(range), [code] "+s"(code) : [v50] "v"(v50) : "scc", "s80", "s81", "s85", "s86", "s87", "s88", "v54");
    } else {
        for (int i = 0; i < iters; i++)
            asm volatile("s_mov_b32 s86, 1024\n s_mov_b32 s89, 0x1000000\n .rept 8\n" VLEVEL ".endr\n"
                : [vr] "+v"(vr), [vc] "+v"(vc), [vm] "+v"(vm) : [v50] "v"(v50) : "vcc", "s86", "s88", "s89", "v43", "v44", "v45", "v46", "v54");
    }
    out[blockIdx.x * 64 + threadIdx.x] = range + code + vr + vc + vm;
}
template <int MODE> double run(int pc, int iters, uint32_t *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * pc), dim3(64), 16120, 0, d, 10, 1u); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * pc), dim3(64), 16120, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 16 * 64 * 4);
    const int iters = 20000;
    const char *names[] = {"scalar flavour", "vector flavour", "mixed (alternating WGs)"};
    for (int mode = 0; mode < 3; mode++)
        for (int pc : {1, 4, 8, 10}) {
            double ms = mode == 0 ? run<0>(pc, iters, d) : mode == 1 ? run<1>(pc, iters, d) : run<2>(pc, iters, d);
            double cyc = ms * 1e-3 * 2.4e9 / (iters * 8.0);
            printf("%-26s per_cu %2d: cycles/level/wave %.1f  cycles/level/CU %.1f\n", names[mode], pc, cyc, cyc / pc);
        }
    return 0;
}
