// How far behind a VALU instruction that writes VCC / an SGPR must the SALU instruction that reads it sit
// (dev tool)?  Every group has the same instructions; only the position of the consumer moves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define F1 "s_add_u32 %1, %1, 7\n"
#define F2 "s_lshr_b32 s81, %1, 3\n"
#define F3 "s_sub_u32 %1, %1, s81\n"
#define F4 "s_xor_b32 %1, %1, s81\n"
#define PROD "v_subrev_co_u32 %3, vcc, %0, %2\n"
#define CONS "s_cmp_lg_u32 vcc_lo, 0\n s_cselect_b32 %0, %0, s80\n s_addc_u32 s80, s80, s80\n"
#define VM "v_min_u32 %2, %2, %3\n"
#define RL "v_readlane_b32 s82, %2, 5\n"
#define RC "s_sub_u32 %0, %0, s82\n s_cselect_b32 %0, %0, s80\n s_addc_u32 s80, s80, s80\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t s = seed, t = seed + 5, v = threadIdx.x + seed, w = v + 1;
    for (int i = 0; i < iters; i++) {
        // VCC path: distance 0..4 scalar instructions, then with the v_min as well
        if (MODE == 0) asm volatile(REP16(PROD CONS VM F1 F2 F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 1) asm volatile(REP16(PROD F1 CONS VM F2 F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 2) asm volatile(REP16(PROD F1 F2 CONS VM F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 3) asm volatile(REP16(PROD F1 F2 F3 CONS VM F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 4) asm volatile(REP16(PROD F1 F2 F3 F4 CONS VM) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 5) asm volatile(REP16(PROD VM F1 F2 F3 F4 CONS) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        // lane read path: v_readlane writes an SGPR, an SALU instruction reads it
        if (MODE == 6) asm volatile(REP16(RL RC VM F1 F2 F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 7) asm volatile(REP16(RL F1 RC VM F2 F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 8) asm volatile(REP16(RL F1 F2 RC VM F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        if (MODE == 9) asm volatile(REP16(RL F1 F2 F3 F4 RC VM) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        // no cross-pipe dependence at all (the consumer reads an SGPR instead of VCC)
        if (MODE == 10) asm volatile(REP16(PROD "s_cmp_lg_u32 s81, 0\n s_cselect_b32 %0, %0, s80\n s_addc_u32 s80, s80, s80\n" VM F1 F2 F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
        // the consumer is a branch on VCC (never taken)
        if (MODE == 11) asm volatile(REP16("v_cmp_eq_u32 vcc, 0x12345, %2\n s_cbranch_vccnz 1f\n1:\n s_cselect_b32 %0, %0, s80\n s_addc_u32 s80, s80, s80\n" VM F1 F2 F3 F4) : "+s"(s), "+s"(t), "+v"(v), "+v"(w)::"scc", "vcc", "s80", "s81", "s82");
    }
    out[blockIdx.x * 64 + threadIdx.x] = s + v + t + w;
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
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 20000;
    const char *names[] = {"vcc: consumer right behind", "vcc: 1 SALU between", "vcc: 2 between", "vcc: 3 between", "vcc: 4 between",
                           "vcc: v_min + 4 between", "readlane: consumer right behind", "readlane: 1 between", "readlane: 2 between",
                           "readlane: 4 between", "no cross-pipe dependence", "branch on vcc right behind v_cmp"};
    for (int mode = 0; mode < 12; mode++) {
        for (int pc : {8, 16}) {
            double ms = 0;
            switch (mode) {
            case 0: ms = run<0>(pc, iters, d); break; case 1: ms = run<1>(pc, iters, d); break;
            case 2: ms = run<2>(pc, iters, d); break; case 3: ms = run<3>(pc, iters, d); break;
            case 4: ms = run<4>(pc, iters, d); break; case 5: ms = run<5>(pc, iters, d); break;
            case 6: ms = run<6>(pc, iters, d); break; case 7: ms = run<7>(pc, iters, d); break;
            case 8: ms = run<8>(pc, iters, d); break; case 9: ms = run<9>(pc, iters, d); break;
            case 10: ms = run<10>(pc, iters, d); break; case 11: ms = run<11>(pc, iters, d); break; }
            double groups = (double)iters * 16 * pc;
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-36s waves/CU %2d: %7.2f ms  cycles/group/wave %6.1f  groups/cycle/CU %.4f (7 SALU + 2 VALU each)\n", names[mode], pc, ms,
                   cyc / ((double)iters * 16), groups / cyc);
        }
    }
    return 0;
}
