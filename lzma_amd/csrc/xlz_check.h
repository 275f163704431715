// xlz_check.h -- integrity checks of the container front-ends (.xz: CRC32 / CRC64 / SHA-256 per
// block; .7z: CRC32 per folder or file).  Host only; the reference has no container code.
#pragma once
#include <cstdint>
#include <cstring>
#include <mutex>

namespace xlzcheck {

inline uint32_t crc32_tab[8][256];
inline uint64_t crc64_tab[4][256];
inline std::once_flag crc_once;

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
        for (int t = 1; t < 8; t++) crc32_tab[t][i] = (crc32_tab[t - 1][i] >> 8) ^ crc32_tab[0][crc32_tab[t - 1][i] & 0xFF];
        for (int t = 1; t < 4; t++) crc64_tab[t][i] = (crc64_tab[t - 1][i] >> 8) ^ crc64_tab[0][crc64_tab[t - 1][i] & 0xFF];
    }
}

inline uint32_t crc32(const uint8_t *p, size_t n)
{
    std::call_once(crc_once, crc_init);
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) { // slicing-by-8
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = crc32_tab[7][a & 0xFF] ^ crc32_tab[6][(a >> 8) & 0xFF] ^ crc32_tab[5][(a >> 16) & 0xFF] ^ crc32_tab[4][a >> 24] ^
            crc32_tab[3][b & 0xFF] ^ crc32_tab[2][(b >> 8) & 0xFF] ^ crc32_tab[1][(b >> 16) & 0xFF] ^ crc32_tab[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ crc32_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

inline uint64_t crc64(const uint8_t *p, size_t n)
{
    std::call_once(crc_once, crc_init);
    uint64_t c = ~0ull;
    while (n >= 4) { // slicing-by-4
        uint32_t a;
        memcpy(&a, p, 4);
        a ^= (uint32_t)c;
        c = (c >> 32) ^ crc64_tab[3][a & 0xFF] ^ crc64_tab[2][(a >> 8) & 0xFF] ^ crc64_tab[1][(a >> 16) & 0xFF] ^
            crc64_tab[0][a >> 24];
        p += 4;
        n -= 4;
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
