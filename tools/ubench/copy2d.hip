// copy2d.hip -- dev probe (GPU box): how does this stack move STRIDED slices between HBM and pinned host memory,
// alone and next to a persistent grid that looks like the decode launch (16 single-wave workgroups per CU, 7.5 KiB of
// LDS each, raised wave priority)?  Decides how a one-round xlz_decode_batch downloads slice k-1 while slice k decodes:
//   (a) ONE hipMemcpy2DAsync per slice (rows = the streams' regions, width = the slice),
//   (b) one hipMemcpyAsync per stream and slice,
//   (c) a gather kernel (rows -> one packed staging range in HBM) + one linear copy.
// usage: copy2d [rows=4096] [row KiB=1024] [slice KiB=256]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                       \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__global__ __launch_bounds__(64) void busy(uint32_t *sink, uint64_t ticks)
{
    extern __shared__ uint32_t lds[];
    __builtin_amdgcn_s_setprio(3);
    const uint64_t t0 = wall_clock64();
    uint32_t a = threadIdx.x, s = blockIdx.x;
    lds[threadIdx.x] = a;
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < 256; i++) {
            a = a * 1664525u + 1013904223u;
            s = __builtin_amdgcn_readfirstlane(a) + (s >> 1);
        }
        lds[(a >> 8) & 63] = s;
    }
    if (a == 0x12345u) sink[0] = s + lds[3];
}

// rows of `width` bytes at stride `pitch` -> packed
__global__ __launch_bounds__(256) void gather_rows(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t pitch,
                                                   size_t width, size_t rows)
{
    const size_t per_row = width / 16;
    const size_t total = per_row * rows;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / per_row, c = i % per_row;
        ((uint4 *)dst)[i] = *(const uint4 *)(src + r * pitch + c * 16);
    }
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const size_t rows = argc > 1 ? atol(argv[1]) : 4096;
    const size_t pitch = (argc > 2 ? atol(argv[2]) : 1024) * 1024 + 256;
    const size_t width = (argc > 3 ? atol(argv[3]) : 256) * 1024;
    uint8_t *dev, *pin, *pack;
    uint32_t *sink;
    CK(hipMalloc(&dev, rows * pitch));
    CK(hipMalloc(&pack, rows * width));
    CK(hipMalloc(&sink, 256));
    CK(hipHostMalloc(&pin, rows * width, hipHostMallocDefault));
    CK(hipMemset(dev, 5, rows * pitch));
    memset(pin, 1, rows * width);
    hipStream_t sc, sk, sg;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    hipEvent_t e0, e1, k0, k1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&k0));
    CK(hipEventCreate(&k1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double gb = rows * width / 1e9;
    printf("%zu rows, pitch %zu, slice %zu bytes = %.3f GB per slice; %d CUs\n", rows, pitch, width, gb, cus);

    auto run = [&](const char *name, bool with_busy, auto &&issue) {
        for (int rep = 0; rep < 2; rep++) {
            if (with_busy) {
                CK(hipEventRecord(k0, sk));
                hipLaunchKernelGGL(busy, dim3(cus * 16), dim3(64), 7680, sk, sink, (uint64_t)(60e-3 * 1e8)); // 60 ms
                CK(hipEventRecord(k1, sk));
            }
            const double t0 = now_ms();
            CK(hipEventRecord(e0, sc));
            issue();
            CK(hipEventRecord(e1, sc));
            const double t_issue = now_ms() - t0;
            CK(hipStreamSynchronize(sc));
            CK(hipStreamSynchronize(sg));
            const double t_done = now_ms() - t0;
            float ms = 0, kms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (with_busy) {
                CK(hipStreamSynchronize(sk));
                CK(hipEventElapsedTime(&kms, k0, k1));
            }
            if (rep == 1)
                printf("%-44s %s: events %7.2f ms = %6.1f GB/s; issue %6.2f ms, done %7.2f ms%s\n", name,
                       with_busy ? "under busy grid" : "alone          ", ms, gb / ms * 1e3, t_issue, t_done,
                       with_busy ? (kms > 62 ? "  [busy kernel STRETCHED]" : "") : "");
            if (with_busy && rep == 1) printf("    busy kernel %.2f ms (60 asked)\n", kms);
        }
    };
    for (int with_busy = 0; with_busy < 2; with_busy++) {
        run("D2H 2D (one call)", with_busy, [&] {
            CK(hipMemcpy2DAsync(pin, width, dev, pitch, width, rows, hipMemcpyDeviceToHost, sc));
        });
        run("H2D 2D (one call)", with_busy, [&] {
            CK(hipMemcpy2DAsync(dev, pitch, pin, width, width, rows, hipMemcpyHostToDevice, sc));
        });
        run("D2H one hipMemcpyAsync per row", with_busy, [&] {
            for (size_t r = 0; r < rows; r++)
                CK(hipMemcpyAsync(pin + r * width, dev + r * pitch, width, hipMemcpyDeviceToHost, sc));
        });
        run("H2D one hipMemcpyAsync per row", with_busy, [&] {
            for (size_t r = 0; r < rows; r++)
                CK(hipMemcpyAsync(dev + r * pitch, pin + r * width, width, hipMemcpyHostToDevice, sc));
        });
        run("D2H gather kernel + one linear copy", with_busy, [&] {
            hipLaunchKernelGGL(gather_rows, dim3(cus * 2), dim3(256), 0, sc, dev, pack, pitch, width, rows);
            CK(hipMemcpyAsync(pin, pack, rows * width, hipMemcpyDeviceToHost, sc));
        });
        run("D2H linear (the whole slice volume)", with_busy, [&] {
            CK(hipMemcpyAsync(pin, pack, rows * width, hipMemcpyDeviceToHost, sc));
        });
        run("H2D linear", with_busy, [&] {
            CK(hipMemcpyAsync(pack, pin, rows * width, hipMemcpyHostToDevice, sc));
        });
    }
    // full duplex: H2D and D2H linear at once
    {
        CK(hipEventRecord(e0, sc));
        CK(hipMemcpyAsync(pin, pack, rows * width / 2, hipMemcpyDeviceToHost, sc));
        CK(hipMemcpyAsync(dev, pin + rows * width / 2, rows * width / 2, hipMemcpyHostToDevice, sg));
        CK(hipStreamSynchronize(sg));
        CK(hipEventRecord(e1, sc));
        CK(hipStreamSynchronize(sc));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("duplex: %.3f GB each way at once: %.2f ms = %.1f GB/s per direction\n", gb / 2, ms, gb / 2 / ms * 1e3);
    }
    return 0;
}
