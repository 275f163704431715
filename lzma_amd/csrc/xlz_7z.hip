// xlz_7z.hip -- .7z container front-end: folder list -> batch of raw LZMA / LZMA2 streams.
//
// SURVEY.md section 8(f) rank 3.  The reference's only real plugin API are the two bodgit/sevenzip
// decompressor constructors (reader1.go:28-61 method id 03 01 01, reader2.go:45-75 method id 21):
// a .7z archive stores its data in "folders", each one compressed stream with out-of-band
// properties -- exactly what those constructors take (props byte + LE32 dictionary size +
// unpackSize for LZMA; one dictionary byte for LZMA2).  Folders are independent, so an archive is
// ONE call of the batch engine: folder i = stream i (XLZ_FMT_LZMA_RAW / XLZ_FMT_LZMA2_RAW).
//
// Host-only code (no kernels here).  Format restated from 7-Zip's published 7zFormat.txt; nothing
// of it exists in the reference.  Only what feeds the LZMA paths is implemented: folders with ONE
// coder that is LZMA, LZMA2 or Copy and one packed stream.  Coder chains (BCJ + LZMA, ...),
// encryption and multi-volume archives are reported per folder as unsupported.  File names and
// attributes (FilesInfo) are not parsed: the output is the folders' bytes back to back, which is
// the archive's files back to back.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/xlz.h"
#include "xlz_check.h"

namespace {

enum : uint8_t {
    kEnd = 0x00, kHeader = 0x01, kArchiveProperties = 0x02, kAdditionalStreamsInfo = 0x03, kMainStreamsInfo = 0x04,
    kFilesInfo = 0x05, kPackInfo = 0x06, kUnPackInfo = 0x07, kSubStreamsInfo = 0x08, kSize = 0x09, kCRC = 0x0A,
    kFolder = 0x0B, kCodersUnPackSize = 0x0C, kNumUnPackStream = 0x0D, kEncodedHeader = 0x17
};

const uint8_t kMagic[6] = {'7', 'z', 0xBC, 0xAF, 0x27, 0x1C};
constexpr uint64_t kMaxItems = 1u << 24; // folders / streams / files we are willing to index

uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }

// bounded reader over a header
struct Rd {
    const uint8_t *p;
    size_t n, pos = 0;
    bool bad = false;
    size_t left() const { return n - pos; }
    uint8_t byte()
    {
        if (pos >= n) {
            bad = true;
            return 0;
        }
        return p[pos++];
    }
    // 7z "NUMBER": the count of leading one bits of the first byte = extra bytes (little endian)
    uint64_t number()
    {
        const uint8_t first = byte();
        uint8_t mask = 0x80;
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) {
            if (!(first & mask)) {
                v |= (uint64_t)(first & (mask - 1)) << (8 * i);
                return v;
            }
            v |= (uint64_t)byte() << (8 * i);
            mask >>= 1;
        }
        return v;
    }
    bool skip(uint64_t k)
    {
        if (k > n - pos) {
            bad = true;
            return false;
        }
        pos += (size_t)k;
        return true;
    }
};

struct Digests {
    std::vector<uint8_t> defined;
    std::vector<uint32_t> crc;
};

bool read_digests(Rd &r, uint64_t count, Digests &d)
{
    if (count > kMaxItems || count > 8 * (uint64_t)r.left()) return false; // >= 1 bit of header per digest
    d.defined.assign((size_t)count, 1);
    d.crc.assign((size_t)count, 0);
    if (r.byte() == 0) { // not all defined: a bit vector, MSB first
        uint8_t b = 0, mask = 0;
        for (uint64_t i = 0; i < count; i++) {
            if (!mask) {
                b = r.byte();
                mask = 0x80;
            }
            d.defined[(size_t)i] = (b & mask) != 0;
            mask >>= 1;
        }
    }
    for (uint64_t i = 0; i < count; i++)
        if (d.defined[(size_t)i]) {
            if (r.n - r.pos < 4) return false;
            d.crc[(size_t)i] = le32(r.p + r.pos);
            r.pos += 4;
        }
    return !r.bad;
}

struct Folder {
    uint32_t method = 0; // XLZ_7Z_*
    uint8_t props = 0;
    uint32_t dict_size = 0;
    uint32_t n_pack = 1;    // packed streams it consumes
    uint32_t n_out = 1;     // output streams of its coder graph
    uint64_t unpack_size = 0;
    bool has_crc = false;
    uint32_t crc = 0;
    uint64_t n_sub = 1;
};

struct Streams {
    uint64_t pack_pos = 0;
    std::vector<uint64_t> pack_sizes;
    std::vector<Folder> folders;
    std::vector<xlz_7z_substream> subs; // in folder order
};

// one folder (7zFormat.txt "Folder"): coders, bind pairs, packed stream indices
bool read_folder(Rd &r, Folder &f)
{
    const uint64_t n_coders = r.number();
    if (n_coders == 0 || n_coders > 32) return false;
    uint64_t total_in = 0, total_out = 0;
    uint32_t method = XLZ_7Z_UNSUPPORTED;
    for (uint64_t c = 0; c < n_coders; c++) {
        const uint8_t mb = r.byte();
        if (mb & 0xC0) return false; // reserved / alternative methods
        const unsigned id_size = mb & 0x0F;
        uint8_t id[16] = {0};
        for (unsigned k = 0; k < id_size; k++) id[k] = r.byte();
        uint64_t n_in = 1, n_out = 1;
        if (mb & 0x10) {
            n_in = r.number();
            n_out = r.number();
            // a coder without an input or an output stream does not exist (7-Zip refuses it); with n_in == 0 the
            // folder below would claim zero packed streams and place_folders would index past PackInfo
            if (n_in == 0 || n_out == 0 || n_in > 32 || n_out > 32) return false;
        }
        uint64_t psz = 0;
        size_t ppos = r.pos;
        if (mb & 0x20) {
            psz = r.number();
            ppos = r.pos;
            if (!r.skip(psz)) return false;
        }
        total_in += n_in;
        total_out += n_out;
        if (n_coders == 1 && n_in == 1 && n_out == 1) {
            if (id_size == 3 && id[0] == 0x03 && id[1] == 0x01 && id[2] == 0x01 && psz == 5) { // LZMA, reader1.go:31
                method = XLZ_7Z_LZMA;
                f.props = r.p[ppos];
                f.dict_size = le32(r.p + ppos + 1);
            } else if (id_size == 1 && id[0] == 0x21 && psz == 1) { // LZMA2, reader2.go:48
                method = XLZ_7Z_LZMA2;
                f.props = r.p[ppos];
            } else if (id_size == 1 && id[0] == 0x00) { // Copy
                method = XLZ_7Z_COPY;
            }
        }
    }
    if (total_out == 0) return false;
    const uint64_t n_bind = total_out - 1;
    if (total_in < n_bind) return false;
    for (uint64_t k = 0; k < n_bind; k++) {
        r.number();
        r.number();
    }
    const uint64_t n_packed = total_in - n_bind;
    if (n_packed == 0) return false; // every folder consumes at least one packed stream
    if (n_packed > 1)
        for (uint64_t k = 0; k < n_packed; k++) r.number();
    f.method = method;
    f.n_pack = (uint32_t)n_packed;
    f.n_out = (uint32_t)total_out;
    return !r.bad;
}

int read_streams_info(Rd &r, Streams &s)
{
    uint8_t id = r.byte();
    if (id == kPackInfo) {
        s.pack_pos = r.number();
        const uint64_t n = r.number();
        if (n > kMaxItems) return XLZ_ERR_UNSUPPORTED;
        id = r.byte();
        if (id == kSize) {
            if (n > r.left()) return XLZ_ERR_RESULT; // every size is at least one byte: no allocation the header cannot back
            s.pack_sizes.resize((size_t)n);
            for (auto &v : s.pack_sizes) v = r.number();
            id = r.byte();
        }
        if (id == kCRC) {
            Digests d;
            if (!read_digests(r, n, d)) return XLZ_ERR_RESULT;
            id = r.byte();
        }
        if (id != kEnd || r.bad) return XLZ_ERR_RESULT;
        id = r.byte();
    }
    if (id == kUnPackInfo) {
        if (r.byte() != kFolder) return XLZ_ERR_RESULT;
        const uint64_t nf = r.number();
        if (nf > kMaxItems) return XLZ_ERR_UNSUPPORTED;
        if (r.byte() != 0) return XLZ_ERR_UNSUPPORTED; // external folder definitions
        if (nf > r.left() / 3) return XLZ_ERR_RESULT; // a folder takes >= 2 header bytes + 1 byte of unpack size
        s.folders.resize((size_t)nf);
        for (auto &f : s.folders)
            if (!read_folder(r, f)) return XLZ_ERR_RESULT;
        if (r.byte() != kCodersUnPackSize) return XLZ_ERR_RESULT;
        for (auto &f : s.folders)
            for (uint32_t k = 0; k < f.n_out; k++) {
                const uint64_t v = r.number();
                if (k + 1 == f.n_out || f.n_out == 1) f.unpack_size = v; // single coder: its one output
            }
        id = r.byte();
        if (id == kCRC) {
            Digests d;
            if (!read_digests(r, nf, d)) return XLZ_ERR_RESULT;
            for (size_t i = 0; i < s.folders.size(); i++) {
                s.folders[i].has_crc = d.defined[i];
                s.folders[i].crc = d.crc[i];
            }
            id = r.byte();
        }
        if (id != kEnd || r.bad) return XLZ_ERR_RESULT;
        id = r.byte();
    }
    bool have_sub = false;
    if (id == kSubStreamsInfo) {
        have_sub = true;
        id = r.byte();
        if (id == kNumUnPackStream) {
            for (auto &f : s.folders) {
                f.n_sub = r.number();
                if (f.n_sub > kMaxItems) return XLZ_ERR_UNSUPPORTED;
            }
            id = r.byte();
        }
        uint64_t total_sub = 0, sized = 0;
        for (auto &f : s.folders) {
            total_sub += f.n_sub;
            sized += f.n_sub ? f.n_sub - 1 : 0; // sizes in the header: all but the last of each folder
        }
        if (total_sub > kMaxItems) return XLZ_ERR_UNSUPPORTED;
        // the sizes must be in the header (one byte each at least); without a kSize attribute a folder has at most
        // one stream whose size is known at all
        if (sized > (id == kSize ? (uint64_t)r.left() : 0)) return XLZ_ERR_RESULT;
        s.subs.clear();
        for (auto &f : s.folders) { // sizes: all but the last of each folder, the last is the rest
            uint64_t sum = 0;
            for (uint64_t k = 0; k + 1 < f.n_sub; k++) {
                const uint64_t v = id == kSize ? r.number() : 0;
                if (v > f.unpack_size - sum) return XLZ_ERR_RESULT;
                sum += v;
                s.subs.push_back({v, 0, 0});
            }
            if (f.n_sub) s.subs.push_back({f.unpack_size - sum, 0, 0});
        }
        if (id == kSize) id = r.byte();
        if (id == kCRC) { // digests of the streams whose CRC is not known from the folder
            uint64_t need = 0;
            for (auto &f : s.folders)
                if (!(f.n_sub == 1 && f.has_crc)) need += f.n_sub;
            Digests d;
            if (!read_digests(r, need, d)) return XLZ_ERR_RESULT;
            size_t di = 0, si = 0;
            for (auto &f : s.folders) {
                for (uint64_t k = 0; k < f.n_sub; k++, si++) {
                    if (f.n_sub == 1 && f.has_crc) {
                        s.subs[si].crc = f.crc;
                        s.subs[si].has_crc = 1;
                    } else {
                        s.subs[si].crc = d.crc[di];
                        s.subs[si].has_crc = d.defined[di];
                        di++;
                    }
                }
            }
            id = r.byte();
        } else {
            size_t si = 0;
            for (auto &f : s.folders)
                for (uint64_t k = 0; k < f.n_sub; k++, si++)
                    if (f.n_sub == 1 && f.has_crc) {
                        s.subs[si].crc = f.crc;
                        s.subs[si].has_crc = 1;
                    }
        }
        if (id != kEnd || r.bad) return XLZ_ERR_RESULT;
        id = r.byte();
    }
    if (!have_sub) { // one stream per folder
        s.subs.clear();
        for (auto &f : s.folders) s.subs.push_back({f.unpack_size, f.has_crc ? f.crc : 0u, f.has_crc ? 1u : 0u});
    }
    if (id != kEnd || r.bad) return XLZ_ERR_RESULT;
    return XLZ_OK;
}

// lay the folders of a StreamsInfo out against the file: packed streams follow each other from
// 32 + PackPos
int place_folders(const Streams &s, size_t file_len, std::vector<xlz_7z_folder> &out)
{
    out.clear();
    uint64_t off = 32 + s.pack_pos, uoff = 0;
    size_t pi = 0, si = 0;
    for (const Folder &f : s.folders) {
        xlz_7z_folder o;
        memset(&o, 0, sizeof o);
        o.method = f.method;
        if (f.n_pack != 1) o.method = XLZ_7Z_UNSUPPORTED;
        if (f.n_pack < 1 || pi >= s.pack_sizes.size() || f.n_pack > s.pack_sizes.size() - pi) return XLZ_ERR_RESULT;
        o.pack_off = off;
        o.pack_len = s.pack_sizes[pi];
        for (uint32_t k = 0; k < f.n_pack; k++) {
            const uint64_t sz = s.pack_sizes[pi + k];
            if (off > file_len || sz > file_len - off) return XLZ_ERR_UNEXPECTED_EOF;
            off += sz;
        }
        pi += f.n_pack;
        o.unpack_off = uoff;
        o.unpack_len = f.unpack_size;
        if (f.unpack_size > ~0ull - uoff) return XLZ_ERR_RESULT;
        uoff += f.unpack_size;
        o.props = f.props;
        o.dict_size = f.method == XLZ_7Z_LZMA2 ? xlz_decode_dict_size2(f.props) : f.dict_size;
        if (f.method == XLZ_7Z_LZMA2 && f.props > 40) o.method = XLZ_7Z_UNSUPPORTED;
        if (f.method == XLZ_7Z_LZMA2 && f.props == 40) o.dict_size = 0xFFFFFFFFu;
        o.has_crc = f.has_crc;
        o.crc = f.crc;
        o.first_substream = (uint32_t)si;
        o.n_substreams = (uint32_t)f.n_sub;
        si += (size_t)f.n_sub;
        out.push_back(o);
    }
    return XLZ_OK;
}

// decode the given folders (all of a StreamsInfo) into `out`, verify their CRCs
int decode_folders(xlz_ctx *ctx, const uint8_t *file, const std::vector<xlz_7z_folder> &fo,
                   const std::vector<xlz_7z_substream> &subs, uint8_t *out, int verify, size_t *unverified,
                   xlz_ctx *const *ctxs = nullptr, size_t n_ctx = 0) // ctxs: deal the folders to several GPUs (xlz_decode_batch_multi)
{
    std::vector<xlz_stream_desc> d;
    std::vector<size_t> which;
    uint8_t propbuf[5];
    (void)propbuf;
    for (size_t i = 0; i < fo.size(); i++) {
        const xlz_7z_folder &f = fo[i];
        if (f.method == XLZ_7Z_UNSUPPORTED) return XLZ_ERR_UNSUPPORTED;
        if (f.method == XLZ_7Z_COPY) {
            if (f.pack_len != f.unpack_len) return XLZ_ERR_RESULT;
            memcpy(out + f.unpack_off, file + f.pack_off, (size_t)f.unpack_len);
            continue;
        }
        xlz_stream_desc s;
        memset(&s, 0, sizeof s);
        s.in = file + f.pack_off;
        s.in_len = (size_t)f.pack_len;
        s.out = out + f.unpack_off;
        s.out_cap = (size_t)f.unpack_len;
        if (f.method == XLZ_7Z_LZMA) { // NewLZMADecompressorForSevenZip(props, unpackSize, readers), reader1.go:32-61
            s.format = XLZ_FMT_LZMA_RAW;
            s.props = f.props;
            s.dict_size = f.dict_size; // (the batch engine applies DecodeDictSize's 4096 floor)
            s.unpack_size = f.unpack_len;
        } else { // NewLZMA2DecompressorForSevenZip(props, _, readers), reader2.go:49-75
            s.format = XLZ_FMT_LZMA2_RAW;
            s.dict_size = f.dict_size;
        }
        d.push_back(s);
        which.push_back(i);
    }
    std::vector<xlz_result> r(d.size());
    if (!d.empty()) {
        int st = n_ctx > 1 ? xlz_decode_batch_multi(ctxs, n_ctx, d.data(), d.size(), r.data())
                           : xlz_decode_batch(ctx, d.data(), d.size(), r.data());
        if (st != XLZ_OK) return st;
    }
    for (size_t k = 0; k < d.size(); k++) {
        if (r[k].status < 0) return r[k].status;
        if (r[k].out_len != fo[which[k]].unpack_len) return XLZ_ERR_RESULT; // a folder decodes to exactly its size
    }
    if (verify) {
        size_t nu = 0;
        std::vector<int> bad(fo.size(), 0);
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 16), fo.size()));
        auto work = [&](unsigned t) {
            for (size_t i = t; i < fo.size(); i += nth) {
                const xlz_7z_folder &f = fo[i];
                uint64_t o = f.unpack_off;
                bool any = false;
                for (uint32_t k = 0; k < f.n_substreams; k++) {
                    const xlz_7z_substream &ss = subs[f.first_substream + k];
                    if (ss.has_crc) {
                        any = true;
                        if (xlzcheck::crc32(out + o, (size_t)ss.size) != ss.crc) bad[i] = 1;
                    }
                    o += ss.size;
                }
                if (f.has_crc) {
                    any = true;
                    if (xlzcheck::crc32(out + f.unpack_off, (size_t)f.unpack_len) != f.crc) bad[i] = 1;
                }
                if (!any && !bad[i]) bad[i] = 2;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nth; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (int b : bad) {
            if (b == 1) return XLZ_ERR_RESULT;
            nu += b == 2;
        }
        if (unverified) *unverified = nu;
    }
    return XLZ_OK;
}

// signature header -> the (possibly still encoded) header bytes
int locate_header(const uint8_t *file, size_t len, const uint8_t *&hdr, size_t &hdr_len)
{
    if (len < 32) return XLZ_ERR_UNEXPECTED_EOF;
    if (memcmp(file, kMagic, 6) != 0) return XLZ_ERR_RESULT;
    if (file[6] != 0) return XLZ_ERR_UNSUPPORTED; // major version
    if (xlzcheck::crc32(file + 12, 20) != le32(file + 8)) return XLZ_ERR_RESULT;
    const uint64_t off = le64(file + 12), size = le64(file + 20);
    if (off > len - 32 || size > len - 32 - off) return XLZ_ERR_UNEXPECTED_EOF;
    hdr = file + 32 + off;
    hdr_len = (size_t)size;
    if (size && xlzcheck::crc32(hdr, hdr_len) != le32(file + 28)) return XLZ_ERR_RESULT;
    return XLZ_OK;
}

// the archive's MAIN StreamsInfo.  An encoded header (what 7-Zip writes by default) is itself a
// folder: it is decoded with the batch engine first (needs ctx).
int main_streams(xlz_ctx *ctx, const uint8_t *file, size_t len, Streams &s, std::vector<uint8_t> &decoded_header)
{
    const uint8_t *hdr;
    size_t hl;
    int st = locate_header(file, len, hdr, hl);
    if (st != XLZ_OK) return st;
    if (hl == 0) return XLZ_OK; // empty archive
    for (int depth = 0; depth < 4; depth++) {
        Rd r{hdr, hl};
        const uint8_t id = r.byte();
        if (id == kEncodedHeader) {
            Streams es;
            st = read_streams_info(r, es);
            if (st != XLZ_OK) return st;
            std::vector<xlz_7z_folder> fo;
            st = place_folders(es, len, fo);
            if (st != XLZ_OK) return st;
            if (fo.size() != 1) return XLZ_ERR_UNSUPPORTED;
            if (fo[0].unpack_len > (256u << 20)) return XLZ_ERR_UNSUPPORTED;
            if (!ctx && fo[0].method != XLZ_7Z_COPY) return XLZ_ERR_DEVICE; // decoding the header needs the GPU
            std::vector<uint8_t> next((size_t)fo[0].unpack_len + 1);
            st = decode_folders(ctx, file, fo, es.subs, next.data(), 1, nullptr);
            if (st != XLZ_OK) return st;
            next.resize((size_t)fo[0].unpack_len);
            decoded_header.swap(next);
            hdr = decoded_header.data();
            hl = decoded_header.size();
            continue;
        }
        if (id != kHeader) return XLZ_ERR_RESULT;
        uint8_t t = r.byte();
        if (t == kArchiveProperties) {
            for (;;) {
                const uint8_t pt = r.byte();
                if (pt == kEnd || r.bad) break;
                if (!r.skip(r.number())) return XLZ_ERR_RESULT;
            }
            t = r.byte();
        }
        if (t == kAdditionalStreamsInfo) return XLZ_ERR_UNSUPPORTED;
        if (t == kMainStreamsInfo) return read_streams_info(r, s);
        return r.bad ? XLZ_ERR_RESULT : XLZ_OK; // no streams: only empty files
    }
    return XLZ_ERR_UNSUPPORTED;
}

} // namespace

extern "C" int xlz_7z_index(xlz_ctx *ctx, const uint8_t *file, size_t len, xlz_7z_folder *folders, size_t max_folders,
                            size_t *n_folders, xlz_7z_substream *substreams, size_t max_substreams, size_t *n_substreams,
                            uint64_t *total_unpacked)
{
    if (!file || !n_folders || (!folders && max_folders) || (!substreams && max_substreams)) return XLZ_ERR_BAD_ARG;
    *n_folders = 0;
    if (n_substreams) *n_substreams = 0;
    if (total_unpacked) *total_unpacked = 0;
    Streams s;
    std::vector<uint8_t> dh;
    int st = main_streams(ctx, file, len, s, dh);
    if (st != XLZ_OK) return st;
    std::vector<xlz_7z_folder> fo;
    st = place_folders(s, len, fo);
    if (st != XLZ_OK) return st;
    uint64_t total = 0;
    for (size_t i = 0; i < fo.size(); i++) {
        if (i < max_folders) folders[i] = fo[i];
        total += fo[i].unpack_len;
    }
    for (size_t i = 0; i < s.subs.size() && i < max_substreams; i++) substreams[i] = s.subs[i];
    *n_folders = fo.size();
    if (n_substreams) *n_substreams = s.subs.size();
    if (total_unpacked) *total_unpacked = total;
    if ((max_folders && fo.size() > max_folders) || (max_substreams && s.subs.size() > max_substreams)) return XLZ_ERR_OUT_CAP;
    return XLZ_OK;
}

static int sz_decode(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap,
                     uint64_t *out_len, int verify, size_t *unverified);

extern "C" int xlz_7z_decode(xlz_ctx *ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap, uint64_t *out_len,
                             int verify, size_t *unverified)
{
    return sz_decode(&ctx, 1, file, len, out, out_cap, out_len, verify, unverified);
}

// several contexts (one per GPU): the folders, and the units inside large LZMA2 folders, are dealt by
// xlz_decode_batch_multi; encoded headers are decoded on the first context
extern "C" int xlz_7z_decode_multi(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out,
                                   size_t out_cap, uint64_t *out_len, int verify, size_t *unverified)
{
    if (!ctxs || !n_ctx) return XLZ_ERR_BAD_ARG;
    return sz_decode(ctxs, n_ctx, file, len, out, out_cap, out_len, verify, unverified);
}

static int sz_decode(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap,
                     uint64_t *out_len, int verify, size_t *unverified)
{
    for (size_t c = 0; c < n_ctx; c++)
        if (!ctxs[c]) return XLZ_ERR_BAD_ARG;
    xlz_ctx *ctx = ctxs[0];
    if (!file || (!out && out_cap) || !out_len) return XLZ_ERR_BAD_ARG;
    *out_len = 0;
    if (unverified) *unverified = 0;
    Streams s;
    std::vector<uint8_t> dh;
    int st = main_streams(ctx, file, len, s, dh);
    if (st != XLZ_OK) return st;
    std::vector<xlz_7z_folder> fo;
    st = place_folders(s, len, fo);
    if (st != XLZ_OK) return st;
    uint64_t total = 0;
    for (auto &f : fo) total += f.unpack_len;
    if (total > out_cap) return XLZ_ERR_OUT_CAP;
    st = decode_folders(ctx, file, fo, s.subs, out, verify, unverified, ctxs, n_ctx);
    if (st != XLZ_OK) return st;
    *out_len = total;
    return XLZ_OK;
}
