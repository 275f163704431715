// Host self-test of lzma_amd/csrc/xlz_check.h (CRC32 / CRC64-XZ slicing-by-16, SHA-256): known answers and a bitwise
// restatement on random lengths and alignments.  Built and run by tests/test_check_host.py (g++, no GPU).
#include "xlz_check.h"
#include <chrono>
#include <cstdio>
#include <vector>

static uint32_t crc32_bitwise(const uint8_t *p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
    }
    return ~c;
}

static uint64_t crc64_bitwise(const uint8_t *p, size_t n)
{
    uint64_t c = ~0ull;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xC96C5795D7870F42ull & (0ull - (c & 1)));
    }
    return ~c;
}

int main(int argc, char **argv)
{
    const uint8_t *kat = (const uint8_t *)"123456789";
    if (xlzcheck::crc32(kat, 9) != 0xCBF43926u) return printf("crc32 known answer\n"), 1;
    if (xlzcheck::crc64(kat, 9) != 0x995DC9BBDF1939FAull) return printf("crc64 known answer\n"), 1;
    uint8_t dg[32];
    xlzcheck::sha256((const uint8_t *)"abc", 3, dg);
    if (dg[0] != 0xba || dg[1] != 0x78 || dg[31] != 0xad) return printf("sha256 known answer\n"), 1;
    std::vector<uint8_t> buf(70000);
    uint64_t x = 88172645463325252ull;
    for (auto &b : buf) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        b = (uint8_t)x;
    }
    const bool have_clmul = xlzcheck::crc_clmul; // (set by the first call above)
    for (int pass = 0; pass < 2; pass++) {       // pass 0: the tables alone; pass 1: carry-less-multiply folding where the CPU has it
        xlzcheck::crc_clmul = pass == 1 && have_clmul;
        for (size_t off = 0; off < 19; off++)
            for (size_t n : {size_t(0), size_t(1), size_t(15), size_t(16), size_t(17), size_t(31), size_t(33), size_t(63), size_t(64),
                             size_t(65), size_t(79), size_t(80), size_t(255), size_t(4097), size_t(65521)}) {
                if (xlzcheck::crc32(buf.data() + off, n) != crc32_bitwise(buf.data() + off, n))
                    return printf("crc32 pass %d off %zu n %zu\n", pass, off, n), 1;
                if (xlzcheck::crc64(buf.data() + off, n) != crc64_bitwise(buf.data() + off, n))
                    return printf("crc64 pass %d off %zu n %zu\n", pass, off, n), 1;
            }
    }
    printf("clmul folding: %s\n", have_clmul ? "used" : "not available on this CPU (tables only)");
    if (argc > 1) { // throughput of one thread
        std::vector<uint8_t> big(64u << 20, 0x5a);
        auto t0 = std::chrono::steady_clock::now();
        const uint64_t a = xlzcheck::crc64(big.data(), big.size());
        auto t1 = std::chrono::steady_clock::now();
        const uint32_t b = xlzcheck::crc32(big.data(), big.size());
        auto t2 = std::chrono::steady_clock::now();
        printf("crc64 %.2f GB/s  crc32 %.2f GB/s  (%llx %x)\n", big.size() / 1e9 / std::chrono::duration<double>(t1 - t0).count(),
               big.size() / 1e9 / std::chrono::duration<double>(t2 - t1).count(), (unsigned long long)a, b);
    }
    printf("ok\n");
    return 0;
}
