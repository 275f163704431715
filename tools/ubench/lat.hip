// Microbenchmarks: single-wave dependent-chain latencies on gfx950 (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k_lat(uint64_t *out, uint32_t seed)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (i * 7 + 3) & 4095;
    __syncthreads();
    uint64_t t0, t1;
    uint32_t s = seed, v = seed + threadIdx.x * 0, a = 0;
    // 0: empty
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 1: dependent SALU chain x64
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP64("s_add_u32 %0, %0, 7\n") : "+s"(s)::"scc");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 2: dependent VALU chain x64
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP64("v_add_u32 %0, %0, 7\n") : "+v"(v));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 3: independent VALU x64 (4 chains)
    uint32_t v1 = v + 1, v2 = v + 2, v3 = v + 3;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("v_add_u32 %0, %0, 7\n v_add_u32 %1, %1, 7\n v_add_u32 %2, %2, 7\n v_add_u32 %3, %3, 7\n")
                 : "+v"(v), "+v"(v1), "+v"(v2), "+v"(v3));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 4: LDS pointer chase x16: ds_read_b32 -> wait -> use as address
    a = (threadIdx.x * 0 + seed) & 4095;
    uint32_t addr = a * 4;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0\n") : "+v"(addr)::"memory");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 5: LDS read -> readfirstlane -> SALU -> v_mov -> address (x16)
    uint32_t sa = (seed & 4095) * 4, tv = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("v_mov_b32 %1, %0\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %0, %1\n s_lshl_b32 %0, %0, 2\n")
                 : "+s"(sa), "+v"(tv)::"memory", "scc");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // 6: v_cmp -> s_cbranch_vccz (not taken) x16 with dependent v_add
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("v_add_u32 %0, %0, 1\n v_cmp_eq_u32 vcc, 0x7fffffff, %0\n s_cbranch_vccnz 1f\n1:\n") : "+v"(v)::"vcc");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[6] = t1 - t0;
    // 7: SALU s_cmp -> s_cbranch_scc (not taken) x16 dependent
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("s_add_u32 %0, %0, 1\n s_cmp_eq_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n1:\n") : "+s"(s)::"scc");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[7] = t1 - t0;
    // 8: taken scalar branches x16
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("s_branch 1f\n s_nop 0\n s_nop 0\n1:\n") :::"scc");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[8] = t1 - t0;
    // 9: v_readfirstlane -> v_mov roundtrip x16 (VALU->SALU->VALU)
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("v_readfirstlane_b32 %0, %1\n v_mov_b32 %1, %0\n") : "+s"(s), "+v"(v));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[9] = t1 - t0;
    // 10: dependent s_mul_i32 x64
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP64("s_mul_i32 %0, %0, 3\n") : "+s"(s));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[10] = t1 - t0;
    // 11: dependent v_mul_u32_u24 x64
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP64("v_mul_u32_u24 %0, %0, 3\n") : "+v"(v));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[11] = t1 - t0;
    // 12: ds_write_b16 + ds_read_u16 same address roundtrip x16
    uint32_t la = 64, lv = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("ds_write_b16 %0, %1\n ds_read_u16 %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, 1\n") : "+v"(la), "+v"(lv)::"memory");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[12] = t1 - t0;
    // 13: s_load_dword dependent chain from global (K$ hit) x16
    uint64_t gp = (uint64_t)(out + 64);
    uint32_t sl = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(REP16("s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_and_b32 %0, %0, 0\n") : "+s"(sl) : "s"(gp) : "memory", "scc");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[13] = t1 - t0;
    if (threadIdx.x == 0) out[20] = s + v + v1 + v2 + v3 + addr + sa + tv + lv + sl;
}

int main()
{
    uint64_t *d, h[32];
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, d, 12345u);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"empty(stamp)", "SALU dep x64", "VALU dep x64", "VALU indep x64", "LDS chase x16 (read+lshl)",
                           "LDS->rfl->salu->vmov x16", "v_add+v_cmp+vccbranch x16", "s_add+s_cmp+sccbranch x16",
                           "taken s_branch x16", "rfl+v_mov x16", "s_mul dep x64", "v_mul_u24 dep x64",
                           "ds_write+ds_read+add x16", "s_load chain x16"};
    int cnt[] = {1, 64, 64, 64, 16, 16, 16, 16, 16, 16, 64, 64, 16, 16};
    for (int i = 0; i < 14; i++)
        printf("%-32s total %6llu  per-iter %.1f\n", names[i], (unsigned long long)h[i],
               (double)((long long)h[i] - (long long)h[0]) / cnt[i]);
    return 0;
}
