"""Test infrastructure: a minimal .7z READER in plain Python (7-Zip's published 7zFormat.txt restated), independent of the
product's parser (lzma_amd/csrc/xlz_7z.hip).  It exists so that archives written by an INDEPENDENT writer -- libarchive,
through `cmake -E tar cf x.7z --format=7zip` -- can be checked on the CPU: start header, an LZMA-encoded header (decoded
with liblzma), pack sizes, folders of ONE coder, sub-stream sizes and CRCs.  Nothing here is shipped or imported by
lzma_amd."""
import lzma
import struct
import zlib

K_END, K_HEADER, K_MAIN_STREAMS, K_FILES, K_PACK_INFO, K_UNPACK_INFO, K_SUBSTREAMS = 0, 1, 4, 5, 6, 7, 8
K_SIZE, K_CRC, K_FOLDER, K_CODERS_UNPACK_SIZE, K_NUM_UNPACK_STREAM, K_ENCODED_HEADER = 9, 10, 11, 12, 13, 0x17


class Cursor:
    def __init__(self, data, at=0):
        self.d, self.at = data, at

    def byte(self):
        self.at += 1
        return self.d[self.at - 1]

    def take(self, n):
        self.at += n
        assert self.at <= len(self.d)
        return self.d[self.at - n:self.at]

    def number(self):
        """7z NUMBER: the leading one bits of the first byte count the extra little-endian bytes"""
        first = self.byte()
        mask, value = 0x80, 0
        for i in range(8):
            if not first & mask:
                return value | ((first & (mask - 1)) << (8 * i))
            value |= self.byte() << (8 * i)
            mask >>= 1
        return value

    def defined_vector(self, n):
        if self.byte():  # all defined
            return [True] * n
        bits = self.take((n + 7) // 8)
        return [bool(bits[i // 8] & (0x80 >> (i % 8))) for i in range(n)]


def streams_info(c):
    """-> dict(pack_pos, pack_sizes, folders=[dict(method, props, unpack_size, crc)], substreams=[[(size, crc)] per folder])"""
    info = {"pack_pos": 0, "pack_sizes": [], "folders": [], "substreams": None}
    t = c.byte()
    if t == K_PACK_INFO:
        info["pack_pos"] = c.number()
        n = c.number()
        t = c.byte()
        if t == K_SIZE:
            info["pack_sizes"] = [c.number() for _ in range(n)]
            t = c.byte()
        if t == K_CRC:
            for d in c.defined_vector(n):
                if d:
                    c.take(4)
            t = c.byte()
        assert t == K_END
        t = c.byte()
    if t == K_UNPACK_INFO:
        assert c.byte() == K_FOLDER
        nf = c.number()
        assert c.byte() == 0  # not external
        for _ in range(nf):
            assert c.number() == 1, "one coder per folder"
            flags = c.byte()
            method = c.take(flags & 15)
            assert not flags & 0x10, "a simple coder"
            props = c.take(c.number()) if flags & 0x20 else b""
            info["folders"].append({"method": bytes(method), "props": bytes(props), "unpack_size": None, "crc": None})
        assert c.byte() == K_CODERS_UNPACK_SIZE
        for f in info["folders"]:
            f["unpack_size"] = c.number()
        t = c.byte()
        if t == K_CRC:
            for f, d in zip(info["folders"], c.defined_vector(nf)):
                if d:
                    f["crc"] = struct.unpack("<I", c.take(4))[0]
            t = c.byte()
        assert t == K_END
        t = c.byte()
    if t == K_SUBSTREAMS:
        counts = [1] * len(info["folders"])
        t = c.byte()
        if t == K_NUM_UNPACK_STREAM:
            counts = [c.number() for _ in info["folders"]]
            t = c.byte()
        sizes = []
        for f, n in zip(info["folders"], counts):
            s = []
            if t == K_SIZE:
                s = [c.number() for _ in range(n - 1)]
            if n:
                s.append(f["unpack_size"] - sum(s))
            sizes.append(s)
        if t == K_SIZE:
            t = c.byte()
        crcs = [[None] * n for n in counts]
        if t == K_CRC:
            need = [(i, j) for i, (f, n) in enumerate(zip(info["folders"], counts)) for j in range(n) if not (n == 1 and f["crc"] is not None)]
            for (i, j), d in zip(need, c.defined_vector(len(need))):
                if d:
                    crcs[i][j] = struct.unpack("<I", c.take(4))[0]
            t = c.byte()
        for i, (f, n) in enumerate(zip(info["folders"], counts)):
            if n == 1 and f["crc"] is not None:
                crcs[i][0] = f["crc"]
        assert t == K_END
        info["substreams"] = [list(zip(s, cr)) for s, cr in zip(sizes, crcs)]
        t = c.byte()
    assert t == K_END
    return info


def lzma1_decode(packed, props, unpack_size):
    d = props[0]
    lc, lp, pb = d % 9, (d // 9) % 5, d // 45
    filt = [{"id": lzma.FILTER_LZMA1, "dict_size": max(struct.unpack("<I", props[1:5])[0], 4096), "lc": lc, "lp": lp, "pb": pb}]
    out = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=filt).decompress(packed, unpack_size)
    assert len(out) == unpack_size
    return out


def read(data):
    """-> dict(plain_header, encoded (bool), main (streams_info of the header), body_start=32): the archive's main streams, the
    header decoded with liblzma when it is an encoded one"""
    assert data[:6] == b"7z\xbc\xaf\x27\x1c"
    assert zlib.crc32(data[12:32]) == struct.unpack("<I", data[8:12])[0]
    off, size, crc = struct.unpack("<QQI", data[12:32])
    nh = data[32 + off:32 + off + size]
    assert zlib.crc32(nh) == crc
    encoded = nh[0] == K_ENCODED_HEADER
    if encoded:
        hi = streams_info(Cursor(nh, 1))
        f = hi["folders"][0]
        assert f["method"] == b"\x03\x01\x01" and len(hi["folders"]) == 1
        packed = data[32 + hi["pack_pos"]:32 + hi["pack_pos"] + hi["pack_sizes"][0]]
        nh = lzma1_decode(packed, f["props"], f["unpack_size"])
        if f["crc"] is not None:
            assert zlib.crc32(nh) == f["crc"]
    c = Cursor(nh)
    assert c.byte() == K_HEADER
    t = c.byte()
    if t == 2:  # ArchiveProperties: skip
        while True:
            pt = c.byte()
            if pt == K_END:
                break
            c.take(c.number())
        t = c.byte()
    main = None
    if t == K_MAIN_STREAMS:
        main = streams_info(c)
    return {"plain_header": bytes(nh), "encoded": encoded, "main": main}


def with_plain_header(data):
    """the same archive with its header stored plainly behind the packed streams (what the product's parser takes WITHOUT a
    device: an encoded header is decoded on the GPU like any stream)"""
    r = read(data)
    off = struct.unpack("<Q", data[12:20])[0]
    body = data[32:32 + off]
    start = struct.pack("<QQI", len(body), len(r["plain_header"]), zlib.crc32(r["plain_header"]))
    return data[:8] + struct.pack("<I", zlib.crc32(start)) + start + body + r["plain_header"]
