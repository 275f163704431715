// (level6: the level with s_nop / small reorders at the pipe crossings; generated from level5.hip)
// Marginal cost of each instruction of the literal-tree level (tools/ubench/level4.hip, adopted order): the level
// with ONE instruction deleted (the arithmetic is then no decoder's any more; the issue pattern is what is measured),
// and the full level at other wave counts.  Dev tool for the next round's sequencer model.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define I0 "s_lshr_b32 s80, %0, 11\n"
#define I1 "s_mul_i32 s80, s80, s86\n"
#define I2 "s_sub_u32 s81, %0, s80\n"
#define I3 "v_subrev_co_u32 v28, vcc, s80, %2\n"
#define I4 "v_min_u32 %2, %2, v28\n"
#define I5 "s_cmp_lg_u32 vcc_lo, 0\n"
#define I6 "s_cselect_b32 %0, s80, s81\n"
#define I7 "s_addc_u32 %1, %1, %1\n"
#define I8 "s_lshr_b32 s81, %0, 24\n"
#define I9 "v_readlane_b32 s86, %3, %1\n"
#define IB "s_cbranch_scc0 2f\n 1:\n"
#define STUB "s_branch 3f\n 2:\n s_lshl_b32 %0, %0, 8\n v_perm_b32 %2, %2, %4, %5\n s_branch 1b\n 3:\n"
#define NOSTUB "2:\n"
#define OPS : "+s"(range), "+s"(m), "+v"(code) : "v"(blk), "s"(cur), "v"(sel) : "scc", "vcc", "s80", "s81", "s86", "v28"
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t range = 0xFFFFFFFFu, m = 1, code = 0x12345678u ^ (seed * 2654435761u) ^ (blockIdx.x * 40503u);
    uint32_t blk = 700 + ((threadIdx.x * 37 + seed) % 700);
    uint32_t cur = 0x9E3779B9u * (blockIdx.x + 1), sel = 0x06050400u;
    asm volatile("v_readlane_b32 s86, %0, 1" ::"v"(blk) : "s86");
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(REP8(I0 I1 I2 I3 I4 I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 1) asm volatile(REP8(I0 I1 I2 "s_nop 0\n" I3 I4 I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 2) asm volatile(REP8(I0 I1 I2 "s_nop 0\n" "s_nop 0\n" I3 I4 I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 3) asm volatile(REP8(I0 I1 I2 I3 I4 "s_nop 0\n" I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 4) asm volatile(REP8(I0 I1 I2 I3 I4 "s_nop 0\n" "s_nop 0\n" I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 5) asm volatile(REP8(I0 "s_nop 0\n" I1 I2 I3 I4 I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 6) asm volatile(REP8(I0 I1 I2 I3 I4 I5 I6 I7 I8 I9 IB "s_nop 0\n" STUB) OPS);
        if (MODE == 7) asm volatile(REP8(I1 I2 I3 I4 I5 I6 I7 I8 I0 I9 IB STUB) OPS);
        if (MODE == 8) asm volatile(REP8(I0 I1 "s_nop 0\n" "s_nop 0\n" I3 I2 I4 I5 I6 I7 I8 I9 IB STUB) OPS);
        if (MODE == 9) asm volatile(REP8(I0 I1 I2 I3 I5 I6 I4 I7 I8 I9 IB STUB) OPS);
        if (MODE == 10) asm volatile(REP8(I0 I1 I2 I3 I4 I5 I6 I7 I9 I8 IB STUB) OPS);
        if (MODE == 11) asm volatile(REP8(I0 I1 I2 "s_nop 0\n" I3 I4 "s_nop 0\n" I5 I6 I7 I8 "s_nop 0\n" I9 IB STUB) OPS);
        m = 1;
        cur = cur * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = range + m + code;
}
template <int MODE>
double run(int per_cu, int iters, uint32_t *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), 4096, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), 4096, 0, d, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
#define CASE(M) case M: ms = run<M>(pc, iters, d); break;
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 50000;
    const char *names[] = {"full level", "nop before v_subrev_co", "2 nops before v_subrev_co", "nop before s_cmp_lg", "2 nops before s_cmp_lg", "nop before s_mul", "nop after branch", "s_lshr 11 hoisted before readlane+branch", "s_sub after v_subrev_co + nop", "v_min after s_cselect", "readlane before s_lshr 24", "nops everywhere between S and V"};
    for (int mode = 0; mode < 12; mode++) {
        int pc = 16;
        double ms = 0;
        switch (mode) { CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) }
        printf("%-42s 16 waves/CU: %7.2f ms  %.2f CU cycles per level\n", names[mode], ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8) / 16);
    }
    for (int pc : {16}) {
        double ms = run<0>(pc, iters, d);
        printf("full level               %2d waves/CU: %7.2f ms  %.2f CU cycles per level\n", pc, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8) / pc);
    }
    return 0;
}
