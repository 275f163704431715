// The bit-tree LEVEL instruction mix in a loop, vs waves per CU (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CORE \
    "s_lshr_b32 s80, %[range], 11\n s_mul_i32 s80, s80, s86\n s_sub_u32 s81, %[range], s80\n s_sub_u32 s87, %[code], s80\n" \
    "s_cselect_b32 %[range], s80, s81\n s_cselect_b32 %[code], %[code], s87\n s_cselect_b32 s81, 0x7e1, 0\n s_cselect_b32 s87, 0, 1\n" \
    "s_sub_u32 s81, s86, s81\n v_ashrrev_i32 v63, 5, s81\n v_sub_u32 v63, s86, v63\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    extern __shared__ uint16_t probs[];
    for (int i = threadIdx.x; i < 8000; i += 64) probs[i] = 1024;
    uint32_t range = 0xFFFFFFFF, code = seed * 2654435761u, m = 1;
    uint32_t v58 = 64, v59 = 64 + 2 * threadIdx.x;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) // full level incl. LDS
            asm volatile("s_mov_b32 s86, 1024\n"
                ".rept 8\n"
                "v_lshl_add_u32 v61, %[m], 2, %[vbl]\n ds_read_u16 v62, v61\n" CORE
                "v_lshl_add_u32 v60, %[m], 1, %[vbu]\n ds_write_b16 v60, v63\n s_lshl1_add_u32 %[m], %[m], s87\n"
                "s_lshr_b32 s80, %[range], 24\n s_cbranch_scc0 1f\n1:\n s_or_b32 %[range], %[range], 0x80000000\n"
                "s_waitcnt lgkmcnt(0)\n v_readlane_b32 s86, v62, s87\n s_and_b32 %[m], %[m], 255\n"
                ".endr\n"
                : [range] "+s"(range), [code] "+s"(code), [m] "+s"(m) : [vbl] "v"(v59), [vbu] "v"(v58)
                : "scc", "memory", "s80", "s81", "s86", "s87", "v60", "v61", "v62", "v63");
        if (MODE == 1) // no LDS at all
            asm volatile("s_mov_b32 s86, 1024\n"
                ".rept 8\n"
                "v_lshl_add_u32 v61, %[m], 2, %[vbl]\n" CORE
                "v_lshl_add_u32 v60, %[m], 1, %[vbu]\n s_lshl1_add_u32 %[m], %[m], s87\n"
                "s_lshr_b32 s80, %[range], 24\n s_cbranch_scc0 1f\n1:\n s_or_b32 %[range], %[range], 0x80000000\n"
                "v_readlane_b32 s86, v61, s87\n s_and_b32 %[m], %[m], 255\n s_or_b32 s86, s86, 1024\n s_and_b32 s86, s86, 2047\n"
                ".endr\n"
                : [range] "+s"(range), [code] "+s"(code), [m] "+s"(m) : [vbl] "v"(v59), [vbu] "v"(v58)
                : "scc", "memory", "s80", "s81", "s86", "s87", "v60", "v61", "v62", "v63");
        if (MODE == 2) // no branch
            asm volatile("s_mov_b32 s86, 1024\n"
                ".rept 8\n"
                "v_lshl_add_u32 v61, %[m], 2, %[vbl]\n ds_read_u16 v62, v61\n" CORE
                "v_lshl_add_u32 v60, %[m], 1, %[vbu]\n ds_write_b16 v60, v63\n s_lshl1_add_u32 %[m], %[m], s87\n"
                "s_lshr_b32 s80, %[range], 24\n s_or_b32 %[range], %[range], 0x80000000\n"
                "s_waitcnt lgkmcnt(0)\n v_readlane_b32 s86, v62, s87\n s_and_b32 %[m], %[m], 255\n"
                ".endr\n"
                : [range] "+s"(range), [code] "+s"(code), [m] "+s"(m) : [vbl] "v"(v59), [vbu] "v"(v58)
                : "scc", "memory", "s80", "s81", "s86", "s87", "v60", "v61", "v62", "v63");
    }
    out[blockIdx.x * 64 + threadIdx.x] = range + code + m;
}
template <int MODE> double run(int pc, int iters, uint32_t *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int lds = 16120;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * pc), dim3(64), lds, 0, d, 10, 1u); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * pc), dim3(64), lds, 0, d, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 16 * 64 * 4);
    const int iters = 20000;
    const char *names[] = {"level (LDS+branch)", "level no LDS", "level no branch"};
    for (int mode = 0; mode < 3; mode++)
        for (int pc : {1, 2, 4, 6, 8, 10}) {
            double ms = mode == 0 ? run<0>(pc, iters, d) : mode == 1 ? run<1>(pc, iters, d) : run<2>(pc, iters, d);
            double cyc = ms * 1e-3 * 2.4e9 / (iters * 8.0);
            printf("%-20s per_cu %2d: %.2f ms  cycles/level/wave %.1f  cycles/level/CU %.1f\n", names[mode], pc, ms, cyc, cyc / pc);
        }
    return 0;
}
