// Per-CU issue throughput with W single-wave workgroups per CU (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    uint32_t s = seed, v = threadIdx.x + seed, s2 = seed + 1, v2 = v + 1, a = threadIdx.x * 4;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(REP64("s_add_u32 %0, %0, 7\n") : "+s"(s)::"scc");
        if (MODE == 1) asm volatile(REP64("v_add_u32 %0, %0, 7\n") : "+v"(v));
        if (MODE == 2) asm volatile(REP16("s_add_u32 %0, %0, 7\n v_add_u32 %1, %1, 7\n s_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 3\n") : "+s"(s), "+v"(v)::"scc");
        if (MODE == 3) asm volatile(REP16("s_add_u32 %0, %0, 7\n v_add_u32 %1, %1, 7\n v_add_u32 %1, %1, 3\n v_add_u32 %1, %1, 5\n") : "+s"(s), "+v"(v)::"scc");
        if (MODE == 4) asm volatile(REP16("ds_read_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, 1\n ds_write_b32 %0, %1\n v_add_u32 %1, %1, 1\n") : "+v"(a), "+v"(v)::"memory");
        if (MODE == 5) asm volatile(REP64("s_mul_i32 %0, %0, 7\n") : "+s"(s));
        if (MODE == 6) asm volatile(REP16("s_add_u32 %0, %0, 1\n s_cmp_eq_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n") : "+s"(s)::"scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = s + v + s2 + v2 + a;
}
template <int MODE>
double run(int per_cu, int iters, uint32_t *d, int lds_bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), lds_bytes, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), lds_bytes, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 20000;
    const char *names[] = {"SALU dep", "VALU dep", "S,V,S,V", "S,V,V,V", "LDS r/w chain(5 instr)", "s_mul dep", "s_add,s_cmp,branch-nt,s_add"};
    int per[] = {64, 64, 64, 64, 80, 64, 64};
    for (int mode = 0; mode < 7; mode++) {
        for (int pc : {1, 2, 4, 8, 10, 16}) {
            int lds = 160 * 1024 / pc - 256; if (lds > 65536) lds = 65536; if (pc == 16) lds = 8192;
            double ms = 0;
            switch (mode) {
            case 0: ms = run<0>(pc, iters, d, lds); break; case 1: ms = run<1>(pc, iters, d, lds); break;
            case 2: ms = run<2>(pc, iters, d, lds); break; case 3: ms = run<3>(pc, iters, d, lds); break;
            case 4: ms = run<4>(pc, iters, d, lds); break; case 5: ms = run<5>(pc, iters, d, lds); break;
            case 6: ms = run<6>(pc, iters, d, lds); break; }
            double instr_per_wave = (double)iters * per[mode];
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-28s per_cu %2d: %.2f ms  cycles/instr/wave %.2f  instr/cycle/CU %.3f\n", names[mode], pc, ms,
                   cyc / instr_per_wave, instr_per_wave * pc / cyc);
        }
    }
    return 0;
}
