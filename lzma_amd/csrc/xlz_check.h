// xlz_check.h -- integrity checks of the container front-ends (.xz: CRC32 / CRC64 / SHA-256 per
// block; .7z: CRC32 per folder or file).  Host only; the reference has no container code.
#pragma once
#include <cstdint>
#include <cstring>
#include <mutex>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace xlzcheck {

inline uint32_t crc32_tab[16][256];
inline uint64_t crc64_tab[16][256];
inline std::once_flag crc_once;
// carry-less-multiply folding (x86-64 hosts with PCLMULQDQ): fold constants for one 128-bit accumulator, derived below
inline uint64_t crc32_fold[2], crc64_fold[2];
inline bool crc_clmul = false;

// x^n mod P for a polynomial of degree `deg` given without its leading term, bit e = coefficient of x^e; the result in the
// bit order of a reflected CRC's 64-bit lane (bit j = coefficient of x^(63 - j))
inline uint64_t crc_xn_mod_reflected(unsigned n, uint64_t poly, unsigned deg)
{
    uint64_t r = 1;
    const uint64_t top = 1ull << (deg - 1), mask = deg == 64 ? ~0ull : (1ull << deg) - 1;
    for (unsigned i = 0; i < n; i++) r = ((r << 1) ^ ((r & top) ? poly : 0)) & mask;
    uint64_t k = 0;
    for (unsigned e = 0; e < 64; e++)
        if (r >> e & 1) k |= 1ull << (63 - e);
    return k;
}

inline void crc_init()
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        uint64_t d = i;
        for (int k = 0; k < 8; k++) {
            c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
            d = (d >> 1) ^ (0xC96C5795D7870F42ull & (0ull - (d & 1)));
        }
        crc32_tab[0][i] = c;
        crc64_tab[0][i] = d;
    }
    for (uint32_t i = 0; i < 256; i++) {
        for (int t = 1; t < 16; t++) crc32_tab[t][i] = (crc32_tab[t - 1][i] >> 8) ^ crc32_tab[0][crc32_tab[t - 1][i] & 0xFF];
        for (int t = 1; t < 16; t++) crc64_tab[t][i] = (crc64_tab[t - 1][i] >> 8) ^ crc64_tab[0][crc64_tab[t - 1][i] & 0xFF];
    }
#if defined(__x86_64__)
    // A 128-bit accumulator X = a x^64 + b (a: the earlier eight bytes) moves over the next sixteen bytes as
    // a (x^192 mod P) + b (x^128 mod P); PCLMULQDQ on bit-reflected operands returns the product times x, hence the
    // exponents 191 and 127.  (Checked against a bit-by-bit CRC on every build: tests/c/check_selftest.cpp.)
    crc64_fold[0] = crc_xn_mod_reflected(191, 0x42F0E1EBA9EA3693ull, 64);
    crc64_fold[1] = crc_xn_mod_reflected(127, 0x42F0E1EBA9EA3693ull, 64);
    crc32_fold[0] = crc_xn_mod_reflected(191, 0x04C11DB7ull, 32);
    crc32_fold[1] = crc_xn_mod_reflected(127, 0x04C11DB7ull, 32);
    crc_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
#endif
}

#if defined(__x86_64__)
// folds all whole 16-byte blocks of p[0..n) (n >= 32) into one: `init` = the CRC register, XORed into the first bytes as
// the table algorithm does; -> the 16 bytes that are congruent to everything read, to be run through the tables from a
// zero register; *used = bytes consumed
__attribute__((target("pclmul,sse4.1"))) inline void crc_fold_blocks(const uint8_t *p, size_t n, uint64_t init, const uint64_t fold[2],
                                                                      uint8_t out[16], size_t *used)
{
    const __m128i k = _mm_set_epi64x((long long)fold[1], (long long)fold[0]);
    __m128i x = _mm_xor_si128(_mm_loadu_si128((const __m128i *)p), _mm_set_epi64x(0, (long long)init));
    size_t i = 16;
    for (; i + 16 <= n; i += 16) {
        const __m128i d = _mm_loadu_si128((const __m128i *)(p + i));
        x = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), d);
    }
    _mm_storeu_si128((__m128i *)out, x);
    *used = i;
}
#endif

// Slicing-by-16 (sixteen table lookups per sixteen bytes, no dependency between them): the container front-ends check
// every decoded byte on host threads, so this loop is what a verified .xz / .7z decode waits for after the download.
inline uint32_t crc32(const uint8_t *p, size_t n)
{
    std::call_once(crc_once, crc_init);
    uint32_t c = 0xFFFFFFFFu;
#if defined(__x86_64__)
    uint8_t folded[16];
    if (crc_clmul && n >= 64) {
        size_t used;
        crc_fold_blocks(p, n, c, crc32_fold, folded, &used);
        uint32_t w[4];
        memcpy(w, folded, 16);
        c = 0;
        for (int k = 0; k < 4; k++)
            c ^= crc32_tab[15 - 4 * k][w[k] & 0xFF] ^ crc32_tab[14 - 4 * k][(w[k] >> 8) & 0xFF] ^
                 crc32_tab[13 - 4 * k][(w[k] >> 16) & 0xFF] ^ crc32_tab[12 - 4 * k][w[k] >> 24];
        p += used;
        n -= used;
    }
#endif
    while (n >= 16) {
        uint32_t w[4];
        memcpy(w, p, 16);
        w[0] ^= c;
        c = 0;
        for (int k = 0; k < 4; k++)
            c ^= crc32_tab[15 - 4 * k][w[k] & 0xFF] ^ crc32_tab[14 - 4 * k][(w[k] >> 8) & 0xFF] ^
                 crc32_tab[13 - 4 * k][(w[k] >> 16) & 0xFF] ^ crc32_tab[12 - 4 * k][w[k] >> 24];
        p += 16;
        n -= 16;
    }
    while (n--) c = (c >> 8) ^ crc32_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

inline uint64_t crc64(const uint8_t *p, size_t n)
{
    std::call_once(crc_once, crc_init);
    uint64_t c = ~0ull;
#if defined(__x86_64__)
    uint8_t folded[16];
    if (crc_clmul && n >= 64) {
        size_t used;
        crc_fold_blocks(p, n, c, crc64_fold, folded, &used);
        uint64_t w[2];
        memcpy(w, folded, 16);
        c = 0;
        for (int k = 0; k < 2; k++)
            for (int j = 0; j < 8; j++) c ^= crc64_tab[15 - 8 * k - j][(w[k] >> (8 * j)) & 0xFF];
        p += used;
        n -= used;
    }
#endif
    while (n >= 16) {
        uint64_t w[2];
        memcpy(w, p, 16);
        w[0] ^= c;
        c = 0;
        for (int k = 0; k < 2; k++)
            for (int j = 0; j < 8; j++) c ^= crc64_tab[15 - 8 * k - j][(w[k] >> (8 * j)) & 0xFF];
        p += 16;
        n -= 16;
    }
    while (n--) c = (c >> 8) ^ crc64_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

// SHA-256 (FIPS 180-4), one shot
inline void sha256(const uint8_t *p, size_t n, uint8_t out[32])
{
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    auto rotr = [](uint32_t x, int k) { return (x >> k) | (x << (32 - k)); };
    auto block = [&](const uint8_t *b) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    };
    size_t i = 0;
    for (; i + 64 <= n; i += 64) block(p + i);
    uint8_t tail[128] = {0};
    const size_t r = n - i;
    memcpy(tail, p + i, r);
    tail[r] = 0x80;
    const size_t tl = r + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    block(tail);
    if (tl == 128) block(tail + 64);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(h[k] >> 24);
        out[4 * k + 1] = (uint8_t)(h[k] >> 16);
        out[4 * k + 2] = (uint8_t)(h[k] >> 8);
        out[4 * k + 3] = (uint8_t)h[k];
    }
}

} // namespace xlzcheck
