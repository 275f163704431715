// xlz_xz.hip -- .xz container front-end: block index -> batch of raw LZMA2 streams.
//
// SURVEY.md section 8(f) rank 3.  The reference has no container code (ReadMe.md:6 sends 7z
// users to bodgit/sevenzip); what it offers a container is NewReader2(in, dictSize)
// (reader2.go:26-41) for ONE raw LZMA2 stream.  An .xz file is a list of blocks, each of which is
// exactly such a stream with its own dictionary -- independent by construction -- so the whole
// file is ONE call of the batch engine: block i = stream i, every block further cut into
// dictionary-reset units by scan_lzma2 (xlz_host.hip).
//
// Host-only code (no kernels here).  Format: "The .xz File Format" 1.0.4 (tukaani.org), restated
// from the published specification; nothing of it exists in the reference.  Only what feeds the
// LZMA2 path is implemented: filter chains with anything but a single LZMA2 filter (BCJ, delta)
// are reported as XLZ_ERR_UNSUPPORTED.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/xlz.h"
#include "xlz_check.h"

namespace {

using xlzcheck::crc32;
using xlzcheck::crc64;

uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// variable-length integer (spec 1.2): 7 bits per byte, at most 9 bytes, no trailing zero byte
bool vli(const uint8_t *p, size_t n, size_t &pos, uint64_t &v)
{
    v = 0;
    for (unsigned i = 0; i < 9; i++) {
        if (pos >= n) return false;
        const uint8_t b = p[pos++];
        v |= (uint64_t)(b & 0x7F) << (7 * i);
        if (!(b & 0x80)) return !(b == 0 && i != 0);
    }
    return false;
}

const uint8_t kHeadMagic[6] = {0xFD, '7', 'z', 'X', 'Z', 0x00};
const uint8_t kFootMagic[2] = {'Y', 'Z'};

unsigned check_size(unsigned type) { return type == 0 ? 0 : type <= 3 ? 4 : type <= 6 ? 8 : type <= 9 ? 16 : type <= 12 ? 32 : 64; }

// DecodeDictSize2 of the reference (reader2.go:296-298) stops at 40; the .xz filter flag allows
// 0..40 with 40 = 4 GiB - 1 (spec 5.3.1)
bool xz_dict_size(uint8_t b, uint32_t &d)
{
    if (b > 40) return false;
    d = b == 40 ? 0xFFFFFFFFu : (2u | (b & 1u)) << (b / 2 + 11);
    return true;
}

struct Stream {
    size_t start = 0, end = 0; // [start, end) of the stream inside the file, padding excluded
    size_t index_start = 0;    // blocks live in [start + 12, index_start)
    unsigned check = 0;
    std::vector<std::pair<uint64_t, uint64_t>> records; // (unpadded size, uncompressed size)
};

// one stream, located from its end (footer -> index -> header); spec 2.1
int parse_stream_backwards(const uint8_t *f, size_t end, Stream &s)
{
    if (end < 32) return XLZ_ERR_UNEXPECTED_EOF;
    const uint8_t *ft = f + end - 12;
    if (memcmp(ft + 10, kFootMagic, 2) != 0) return XLZ_ERR_RESULT;
    if (crc32(ft + 4, 6) != le32(ft)) return XLZ_ERR_RESULT;
    if (ft[8] != 0 || (ft[9] & 0xF0)) return XLZ_ERR_UNSUPPORTED;
    s.check = ft[9] & 0x0F;
    const uint64_t index_size = ((uint64_t)le32(ft + 4) + 1) * 4;
    if (index_size + 24 > end) return XLZ_ERR_RESULT;
    const size_t ix = end - 12 - (size_t)index_size; // where the index starts
    s.index_start = ix;
    const uint8_t *ip = f + ix;
    if (crc32(ip, (size_t)index_size - 4) != le32(ip + index_size - 4)) return XLZ_ERR_RESULT;
    if (ip[0] != 0x00) return XLZ_ERR_RESULT;
    size_t pos = 1;
    uint64_t nrec;
    if (!vli(ip, (size_t)index_size - 4, pos, nrec)) return XLZ_ERR_RESULT;
    if (nrec > index_size / 2) return XLZ_ERR_RESULT;
    uint64_t blocks_total = 0;
    s.records.clear();
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t unpadded, uncomp;
        if (!vli(ip, (size_t)index_size - 4, pos, unpadded) || !vli(ip, (size_t)index_size - 4, pos, uncomp)) return XLZ_ERR_RESULT;
        // every block lies between the 12-byte stream header and the index: a record that does not
        // fit what is left there is rejected before it is added (no 64-bit wrap of the sum)
        if (unpadded < 5 || ix < 12 || unpadded > (uint64_t)ix - 12 - blocks_total) return XLZ_ERR_RESULT;
        const uint64_t padded = (unpadded + 3) & ~3ull;
        if (padded > (uint64_t)ix - 12 - blocks_total) return XLZ_ERR_RESULT;
        if (uncomp > (1ull << 62)) return XLZ_ERR_RESULT;
        s.records.emplace_back(unpadded, uncomp);
        blocks_total += padded;
    }
    while (pos < index_size - 4)
        if (ip[pos++] != 0) return XLZ_ERR_RESULT; // index padding
    if (blocks_total + 12 > ix) return XLZ_ERR_RESULT;
    s.start = ix - (size_t)blocks_total - 12;
    s.end = end;
    const uint8_t *hd = f + s.start;
    if (memcmp(hd, kHeadMagic, 6) != 0) return XLZ_ERR_RESULT;
    if (crc32(hd + 6, 2) != le32(hd + 8)) return XLZ_ERR_RESULT;
    if (hd[6] != ft[8] || hd[7] != ft[9]) return XLZ_ERR_RESULT;
    return XLZ_OK;
}

// block header (spec 3.1): only "one LZMA2 filter" chains are accepted
int parse_block_header(const uint8_t *p, size_t avail, size_t &hdr_size, uint32_t &dict, uint64_t &comp_size,
                       uint64_t &uncomp_size)
{
    if (avail < 8 || p[0] == 0) return XLZ_ERR_RESULT;
    hdr_size = ((size_t)p[0] + 1) * 4;
    if (hdr_size > avail) return XLZ_ERR_UNEXPECTED_EOF;
    if (crc32(p, hdr_size - 4) != le32(p + hdr_size - 4)) return XLZ_ERR_RESULT;
    const uint8_t flags = p[1];
    if (flags & 0x3C) return XLZ_ERR_UNSUPPORTED;
    const unsigned nfilters = (flags & 3) + 1;
    size_t pos = 2;
    comp_size = uncomp_size = ~0ull;
    if ((flags & 0x40) && !vli(p, hdr_size - 4, pos, comp_size)) return XLZ_ERR_RESULT;
    if ((flags & 0x80) && !vli(p, hdr_size - 4, pos, uncomp_size)) return XLZ_ERR_RESULT;
    bool have_lzma2 = false;
    for (unsigned k = 0; k < nfilters; k++) {
        uint64_t id, psz;
        if (!vli(p, hdr_size - 4, pos, id) || !vli(p, hdr_size - 4, pos, psz)) return XLZ_ERR_RESULT;
        if (pos + psz > hdr_size - 4) return XLZ_ERR_RESULT;
        if (id == 0x21 && psz == 1 && k + 1 == nfilters) {
            if (!xz_dict_size(p[pos], dict)) return XLZ_ERR_RESULT;
            have_lzma2 = true;
        } else {
            return XLZ_ERR_UNSUPPORTED; // BCJ / delta / anything else in the chain
        }
        pos += (size_t)psz;
    }
    while (pos < hdr_size - 4)
        if (p[pos++] != 0) return XLZ_ERR_RESULT;
    return have_lzma2 && nfilters == 1 ? XLZ_OK : XLZ_ERR_UNSUPPORTED;
}

} // namespace

// Every block of every stream of an .xz file, in file order (spec 2: concatenated streams and
// stream padding allowed).  Pure host parse; no device needed.
extern "C" int xlz_xz_index(const uint8_t *file, size_t len, xlz_xz_block *blocks, size_t max_blocks, size_t *n_blocks,
                            uint64_t *total_uncompressed)
{
    if (!file || !n_blocks || (!blocks && max_blocks)) return XLZ_ERR_BAD_ARG;
    *n_blocks = 0;
    if (total_uncompressed) *total_uncompressed = 0;
    std::vector<Stream> streams;
    size_t end = len;
    while (end > 0) {
        while (end >= 4 && le32(file + end - 4) == 0) end -= 4; // stream padding
        if (end == 0) break;
        if (end & 3) return XLZ_ERR_RESULT;
        Stream s;
        const int st = parse_stream_backwards(file, end, s);
        if (st != XLZ_OK) return st;
        streams.push_back(std::move(s));
        end = streams.back().start;
    }
    if (streams.empty()) return XLZ_ERR_UNEXPECTED_EOF;
    std::reverse(streams.begin(), streams.end());
    uint64_t uoff = 0;
    size_t nb = 0;
    for (const Stream &s : streams) {
        size_t pos = s.start + 12;
        for (const auto &rec : s.records) {
            size_t hdr;
            uint32_t dict = 0;
            uint64_t csz, usz;
            const uint64_t padded = (rec.first + 3) & ~3ull;
            if (pos > s.index_start || padded > s.index_start - pos) return XLZ_ERR_RESULT; // block inside the stream
            const int st = parse_block_header(file + pos, s.index_start - pos, hdr, dict, csz, usz);
            if (st != XLZ_OK) return st;
            const unsigned chk = check_size(s.check);
            if (rec.first < hdr + chk) return XLZ_ERR_RESULT;
            const uint64_t comp = rec.first - hdr - chk;
            if (csz != ~0ull && csz != comp) return XLZ_ERR_RESULT;
            if (usz != ~0ull && usz != rec.second) return XLZ_ERR_RESULT;
            for (uint64_t k = comp; k < ((comp + 3) & ~3ull); k++) // Block Padding: null bytes (liblzma refuses others)
                if (file[pos + hdr + k] != 0) return XLZ_ERR_RESULT;
            if (nb < max_blocks) {
                xlz_xz_block &b = blocks[nb];
                b.comp_off = pos + hdr;
                b.comp_len = comp;
                b.uncomp_off = uoff;
                b.uncomp_len = rec.second;
                b.dict_size = dict;
                b.check_type = s.check;
                b.check_off = pos + hdr + ((comp + 3) & ~3ull);
            }
            nb++;
            if (rec.second > ~0ull - uoff) return XLZ_ERR_RESULT;
            uoff += rec.second;
            pos += (size_t)padded;
        }
    }
    *n_blocks = nb;
    if (total_uncompressed) *total_uncompressed = uoff;
    return nb > max_blocks && max_blocks ? XLZ_ERR_OUT_CAP : XLZ_OK;
}

// Whole file: index, one batch (block = raw LZMA2 stream, reader2.go:26-41), optional integrity
// check of every block (CRC32 / CRC64 on host threads; other check types are left unverified
// and reported through *unverified).
static int xz_decode(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap,
                     uint64_t *out_len, int verify, size_t *unverified);

extern "C" int xlz_xz_decode(xlz_ctx *ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap, uint64_t *out_len,
                             int verify, size_t *unverified)
{
    return xz_decode(&ctx, 1, file, len, out, out_cap, out_len, verify, unverified);
}

// several contexts (one per GPU): the blocks, and the units inside large blocks, are dealt by xlz_decode_batch_multi
extern "C" int xlz_xz_decode_multi(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out,
                                   size_t out_cap, uint64_t *out_len, int verify, size_t *unverified)
{
    if (!ctxs || !n_ctx) return XLZ_ERR_BAD_ARG;
    return xz_decode(ctxs, n_ctx, file, len, out, out_cap, out_len, verify, unverified);
}

static int xz_decode(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap,
                     uint64_t *out_len, int verify, size_t *unverified)
{
    for (size_t c = 0; c < n_ctx; c++)
        if (!ctxs[c]) return XLZ_ERR_BAD_ARG;
    if (!file || (!out && out_cap) || !out_len) return XLZ_ERR_BAD_ARG;
    *out_len = 0;
    if (unverified) *unverified = 0;
    size_t nb = 0;
    uint64_t total = 0;
    int st = xlz_xz_index(file, len, nullptr, 0, &nb, &total);
    if (st != XLZ_OK) return st;
    if (total > out_cap) return XLZ_ERR_OUT_CAP;
    std::vector<xlz_xz_block> blk(nb);
    st = xlz_xz_index(file, len, blk.data(), nb, &nb, &total);
    if (st != XLZ_OK) return st;
    std::vector<xlz_stream_desc> d(nb);
    std::vector<xlz_result> r(nb);
    for (size_t i = 0; i < nb; i++) {
        memset(&d[i], 0, sizeof d[i]);
        d[i].in = file + blk[i].comp_off;
        d[i].in_len = (size_t)blk[i].comp_len;
        d[i].out = out + blk[i].uncomp_off;
        d[i].out_cap = (size_t)blk[i].uncomp_len;
        d[i].format = XLZ_FMT_LZMA2_RAW;
        d[i].dict_size = blk[i].dict_size;
    }
    st = n_ctx > 1 ? xlz_decode_batch_multi(ctxs, n_ctx, d.data(), nb, r.data()) : xlz_decode_batch(ctxs[0], d.data(), nb, r.data());
    if (st != XLZ_OK) return st;
    for (size_t i = 0; i < nb; i++) {
        if (r[i].status < 0) return r[i].status;
        // a block must produce exactly what the index says and use its whole payload
        if (r[i].out_len != blk[i].uncomp_len || r[i].in_consumed != blk[i].comp_len) return XLZ_ERR_RESULT;
    }
    if (verify) {
        std::vector<int> bad(nb, 0);
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned nth = (unsigned)std::min<size_t>(std::max<size_t>(1, std::min<unsigned>(hw ? hw : 1, 16)), std::max<size_t>(nb, 1));
        auto work = [&](unsigned t) {
            for (size_t i = t; i < nb; i += nth) {
                const uint8_t *p = out + blk[i].uncomp_off;
                const uint8_t *c = file + blk[i].check_off;
                if (blk[i].check_type == 1)
                    bad[i] = crc32(p, (size_t)blk[i].uncomp_len) != le32(c);
                else if (blk[i].check_type == 4)
                    bad[i] = crc64(p, (size_t)blk[i].uncomp_len) != ((uint64_t)le32(c) | (uint64_t)le32(c + 4) << 32);
                else if (blk[i].check_type == 10) {
                    uint8_t dg[32];
                    xlzcheck::sha256(p, (size_t)blk[i].uncomp_len, dg);
                    bad[i] = memcmp(dg, c, 32) != 0;
                } else if (blk[i].check_type != 0)
                    bad[i] = 2; // reserved check types: not verified
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nth; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        size_t nu = 0;
        for (size_t i = 0; i < nb; i++) {
            if (bad[i] == 1) return XLZ_ERR_RESULT;
            nu += bad[i] == 2;
        }
        if (unverified) *unverified = nu;
    }
    *out_len = total;
    return XLZ_OK;
}
