// One literal-tree level of the decoder as it is generated (tools/gen_fastpath.py: walk_rec + decide + the hoisted
// lane read), with a live range / code / normalisation stub, in the orders round 2 measured inside the kernel
// (profiles/r02/layout_scan.md section 3).  Question for the next round: does this microbenchmark rank the orders
// the way the kernel does?  If so, orders can be searched here in seconds instead of in A/B builds.
//   0 adopted: s_sub, v_subrev_co, v_min, s_cmp_lg, s_cselect, s_addc, s_lshr, v_readlane, s_cbranch
//   1 order1 : s_cmp_lg in front of v_min                       (kernel: 1.7-2.9 % slower)
//   2 order3 : v_subrev_co in front of s_sub                    (kernel: up to 3.5 % slower)
//   3 no hoist: v_readlane behind the branch                    (kernel: 0.5-2.4 % slower)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define HEAD "s_lshr_b32 s80, %0, 11\n s_mul_i32 s80, s80, s86\n"
#define TAIL_HOIST "s_addc_u32 %1, %1, %1\n s_lshr_b32 s81, %0, 24\n v_readlane_b32 s86, %3, %1\n s_cbranch_scc0 2f\n 1:\n"
#define TAIL_PLAIN "s_addc_u32 %1, %1, %1\n s_lshr_b32 s81, %0, 24\n s_cbranch_scc0 2f\n 1:\n v_readlane_b32 s86, %3, %1\n"
// the stub: range <<= 8, code = code << 8 | byte (v_perm), back
#define STUB "s_branch 3f\n 2:\n s_lshl_b32 %0, %0, 8\n v_perm_b32 %2, %2, %4, %5\n s_branch 1b\n 3:\n"
#define L0 HEAD "s_sub_u32 s81, %0, s80\n v_subrev_co_u32 v28, vcc, s80, %2\n v_min_u32 %2, %2, v28\n s_cmp_lg_u32 vcc_lo, 0\n s_cselect_b32 %0, s80, s81\n" TAIL_HOIST STUB
#define L1 HEAD "s_sub_u32 s81, %0, s80\n v_subrev_co_u32 v28, vcc, s80, %2\n s_cmp_lg_u32 vcc_lo, 0\n v_min_u32 %2, %2, v28\n s_cselect_b32 %0, s80, s81\n" TAIL_HOIST STUB
#define L2 HEAD "v_subrev_co_u32 v28, vcc, s80, %2\n s_sub_u32 s81, %0, s80\n v_min_u32 %2, %2, v28\n s_cmp_lg_u32 vcc_lo, 0\n s_cselect_b32 %0, s80, s81\n" TAIL_HOIST STUB
#define L3 HEAD "s_sub_u32 s81, %0, s80\n v_subrev_co_u32 v28, vcc, s80, %2\n v_min_u32 %2, %2, v28\n s_cmp_lg_u32 vcc_lo, 0\n s_cselect_b32 %0, s80, s81\n" TAIL_PLAIN STUB
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t range = 0xFFFFFFFFu, m = 1, code = 0x12345678u ^ (seed * 2654435761u) ^ (blockIdx.x * 40503u);
    uint32_t blk = 700 + ((threadIdx.x * 37 + seed) % 700); // probabilities 700..1399 of 2048, one per lane
    uint32_t cur = 0x9E3779B9u * (blockIdx.x + 1), sel = 0x06050400u;
    asm volatile("v_readlane_b32 s86, %0, 1" ::"v"(blk) : "s86");
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(REP8(L0) : "+s"(range), "+s"(m), "+v"(code) : "v"(blk), "s"(cur), "v"(sel) : "scc", "vcc", "s80", "s81", "s86", "v28");
        if (MODE == 1) asm volatile(REP8(L1) : "+s"(range), "+s"(m), "+v"(code) : "v"(blk), "s"(cur), "v"(sel) : "scc", "vcc", "s80", "s81", "s86", "v28");
        if (MODE == 2) asm volatile(REP8(L2) : "+s"(range), "+s"(m), "+v"(code) : "v"(blk), "s"(cur), "v"(sel) : "scc", "vcc", "s80", "s81", "s86", "v28");
        if (MODE == 3) asm volatile(REP8(L3) : "+s"(range), "+s"(m), "+v"(code) : "v"(blk), "s"(cur), "v"(sel) : "scc", "vcc", "s80", "s81", "s86", "v28");
        m = 1; // a new literal
        cur = cur * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = range + m + code;
}
template <int MODE>
double run(int per_cu, int iters, uint32_t *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), 8192, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), 8192, 0, d, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 100000;
    const char *names[] = {"adopted order", "order1 (s_cmp_lg before v_min)", "order3 (v_subrev_co before s_sub)", "lane read behind the branch"};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 4; mode++) {
            double ms = mode == 0 ? run<0>(16, iters, d) : mode == 1 ? run<1>(16, iters, d) : mode == 2 ? run<2>(16, iters, d) : run<3>(16, iters, d);
            double levels = (double)iters * 8;
            printf("%-36s 16 waves/CU: %8.2f ms  %.1f cycles per level per wave, %.3f levels per CU cycle\n", names[mode], ms,
                   ms * 1e-3 * 2.4e9 / levels, levels * 16 / (ms * 1e-3 * 2.4e9));
        }
    return 0;
}
