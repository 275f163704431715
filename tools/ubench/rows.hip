// "several decoders per wave on the VALU" measured before it is built.
// MODE 0: today's literal-tree level (tools/ubench/level7.hip, adopted order): ONE decoder per wave, range / slot on the scalar
//         side, 11 instructions per decision.
// MODE 1: FOUR decoders per wave, one per 16-lane DPP row (every lane of a row carries its decoder's range / code / slot):
//         the level entirely on the VALU -- probability fetched from LDS at the row's own address (the four models of the
//         wave live in LDS: 4 x 7416 B, so FIVE such waves fit a CU = the same 20 decoders as today), bound, compare,
//         select of range and code, model update (31 p + c) >> 5 stored back, slot = 2 slot + !bit, and the normalisation
//         as selects (range << 8, code << 8 | next byte of the row's own input word, byte counter) -- no branch at all,
//         i.e. the cheapest possible form: no literal / match / rep divergence, no input refill, no output.
// Printed: cycles per level step per wave and DECISIONS per CU cycle at several wave counts; LDS per workgroup is sized so
// that exactly that many workgroups fit a CU.
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
#define OPS0 : "+s"(range), "+s"(m), "+v"(code) : "v"(blk), "s"(cur), "v"(sel) : "scc", "vcc", "s80", "s81", "s86", "v27", "v28"

// one row-parallel level: %0 range %1 code %2 tree slot %3 model base (LDS byte address
// of the row's tree) %4 input word %5 bytes left in it; v20..v27 temporaries; s70 = 2048, s71 = 1 << 24
#define ROWLEVEL                                                                                                        \
    "v_lshl_add_u32 v20, %2, 1, %3\n ds_read_u16 v21, v20\n v_lshrrev_b32 v22, 11, %0\n s_waitcnt lgkmcnt(0)\n"                  \
    "v_mul_u32_u24 v22, v22, v21\n v_sub_co_u32 v23, vcc, %1, v22\n v_min_u32 %1, %1, v23\n v_sub_u32 v24, %0, v22\n"   \
    "v_cndmask_b32 %0, v24, v22, vcc\n v_mov_b32 v25, 31\n v_cndmask_b32 v25, v25, v26, vcc\n v_mad_u32_u24 v21, v21, 31, v25\n" \
    "v_lshrrev_b32 v21, 5, v21\n ds_write_b16 v20, v21\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n"                            \
    "v_cmp_gt_u32 vcc, s71, %0\n v_lshlrev_b32 v24, 8, %0\n v_perm_b32 v23, %1, %4, v27\n v_cndmask_b32 %0, %0, v24, vcc\n" \
    "v_cndmask_b32 %1, %1, v23, vcc\n v_lshrrev_b32 v24, 8, %4\n v_cndmask_b32 %4, %4, v24, vcc\n v_subb_co_u32 %5, vcc, %5, 0, vcc\n"
#define OPS1 : "+v"(vr), "+v"(vc), "+v"(vm), "+v"(vbase), "+v"(vin), "+v"(vleft) : : "vcc", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"

extern __shared__ uint16_t lds[];
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed, uint32_t lds_probs)
{
    for (uint32_t i = threadIdx.x; i < lds_probs; i += 64) lds[i] = 1024;
    __syncthreads();
    if (MODE == 0) {
        uint32_t range = 0xFFFFFFFFu, m = 1, code = 0x12345678u ^ (seed * 2654435761u) ^ (blockIdx.x * 40503u);
        uint32_t blk = 700 + ((threadIdx.x * 37 + seed) % 700);
        uint32_t cur = 0x9E3779B9u * (blockIdx.x + 1), sel = 0x06050400u;
        asm volatile("v_readlane_b32 s86, %0, 1" ::"v"(blk) : "s86");
        for (int i = 0; i < iters; i++) {
            asm volatile(REP8(I0 I1 I2 I3 I4 I5 I6 I7 I8 I9 IB STUB) OPS0);
            m = 1;
            cur = cur * 1664525u + 1013904223u;
        }
        out[blockIdx.x * 64 + threadIdx.x] = range + m + code;
    } else {
        const uint32_t row = threadIdx.x >> 4;
        uint32_t vr = 0xFFFFFFFFu, vc = (0x12345678u ^ (seed * 2654435761u) ^ (blockIdx.x * 40503u)) + row * 0x9E3779B9u;
        uint32_t vm = 1, vbase = row * 7416u + 3192u, vin = 0x9E3779B9u * (blockIdx.x * 4 + row + 1), vleft = 1u << 30;
        asm volatile("s_movk_i32 s70, 2048\n s_mov_b32 s71, 0x1000000\n v_mov_b32 v26, s70\n v_mov_b32 v27, 0x06050400" ::: "s70", "s71", "v26", "v27");
        for (int i = 0; i < iters; i++) {
            asm volatile(REP8(ROWLEVEL) OPS1);
            vm = 1;   // back to the tree's root; eight levels stay inside the row's model
            vin = vin * 1664525u + 1013904223u;
        }
        out[blockIdx.x * 64 + threadIdx.x] = vr + vc + vm + vleft;
    }
}
template <int MODE>
double run(int per_cu, int iters, uint32_t *d)
{
    // LDS per workgroup so that exactly per_cu workgroups fit the CU's 160 KiB (allocation granule 1280 bytes)
    uint32_t bytes = (160u * 1024u / per_cu) / 1280u * 1280u;
    const uint32_t need = MODE == 0 ? 7416u : 4u * 7416u;
    if (bytes < need) return -1;
    if (bytes > 65536u) bytes = 65536u;
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), bytes, 0, d, 10, 1u, need / 2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * per_cu), dim3(64), bytes, 0, d, iters, 1u, need / 2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 32 * 64 * 4);
    const int iters = 20000;
    for (int pc : {4, 8, 16, 20}) {
        double ms = run<0>(pc, iters, d);
        if (ms < 0) continue;
        double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8);
        printf("one decoder per wave (today's level)     %2d waves/CU: %8.2f ms  %6.1f cycles per level per wave  %.4f decisions per CU cycle\n",
               pc, ms, cyc, pc / cyc);
    }
    for (int pc : {1, 2, 4, 5}) {
        double ms = run<1>(pc, iters, d);
        if (ms < 0) continue;
        double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8);
        printf("four decoders per wave (rows, VALU only) %2d waves/CU: %8.2f ms  %6.1f cycles per level per wave  %.4f decisions per CU cycle (%d decoders per CU)\n",
               pc, ms, cyc, 4.0 * pc / cyc, 4 * pc);
    }
    return 0;
}
