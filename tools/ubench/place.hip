// Where do single-wave workgroups land?  (dev tool)  Histogram of waves per SIMD per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(uint32_t *out, int spin)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t s = 0;
    for (int i = 0; i < spin; i++) asm volatile("s_add_u32 %0, %0, 1\n s_nop 7" : "+s"(s)::"scc");
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = hw; out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = xcc + s * 0; }
}
int main()
{
    uint32_t *d; hipMalloc(&d, 1 << 20);
    for (int wg_waves : {1, 2, 4, 5}) {
        int per_cu_waves = 10;
        if (wg_waves == 4) per_cu_waves = 8;
        int nwg = 256 * per_cu_waves / wg_waves;
        int lds = 160 * 1024 / (per_cu_waves / wg_waves) - 512; if (lds > 65536 && wg_waves < 4) lds = 16000 * wg_waves;
        lds = 16120 * wg_waves;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipMemset(d, 0, 1 << 20);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(64 * wg_waves), lds, 0, d, 20000);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(nwg * wg_waves * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<uint32_t, std::vector<int>> cu; // key: xcc, se, sh, cu -> waves per simd
        for (int i = 0; i < nwg * wg_waves; i++) {
            uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            uint32_t simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cuid;
            auto &v = cu[key]; if (v.empty()) v.assign(4, 0);
            v[simd]++;
        }
        std::map<std::vector<int>, int> hist;
        for (auto &kv : cu) { auto v = kv.second; std::sort(v.begin(), v.end(), std::greater<int>()); hist[v]++; }
        printf("wg = %d waves, %d WGs, LDS %d B: %zu CUs seen. waves-per-SIMD patterns (sorted desc) -> #CUs\n", wg_waves, nwg, lds, cu.size());
        for (auto &kv : hist) printf("   (%d,%d,%d,%d): %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    }
    return 0;
}
