"""Test infrastructure: a minimal .7z WRITER (7-Zip's published 7zFormat.txt restated), used to build
most of the archives the .7z front-end is tested on (test_7z_container.py also writes archives with libarchive, through cmake).  Folders hold
ONE coder each (LZMA, LZMA2 or Copy), which is what the reference's sevenzip constructors take
(reader1.go:28-61, reader2.go:45-75); solid folders with several files, per-file CRCs, plain and
encoded (LZMA-compressed) headers."""
import lzma
import struct
import zlib

K_END, K_HEADER, K_MAIN_STREAMS, K_FILES, K_PACK_INFO, K_UNPACK_INFO, K_SUBSTREAMS = 0, 1, 4, 5, 6, 7, 8
K_SIZE, K_CRC, K_FOLDER, K_CODERS_UNPACK_SIZE, K_NUM_UNPACK_STREAM, K_NAMES, K_ENCODED_HEADER = 9, 10, 11, 12, 13, 0x11, 0x17


def number(v):
    """7z NUMBER: leading one bits of the first byte = number of extra (little-endian) bytes"""
    for n in range(8):
        if v < (1 << (7 * (n + 1))):
            first = ((0xFF << (8 - n)) & 0xFF) | (v >> (8 * n))
            return bytes([first]) + (v & ((1 << (8 * n)) - 1)).to_bytes(n, "little")
    return b"\xff" + v.to_bytes(8, "little")


def lzma_folder(data, dict_size=1 << 16, lc=3, lp=0, pb=2):
    """-> (coder record, packed bytes)   method 03 01 01, props = props byte + LE32 dict size"""
    filt = [{"id": lzma.FILTER_LZMA1, "dict_size": dict_size, "lc": lc, "lp": lp, "pb": pb, "preset": 1}]
    packed = lzma.compress(data, format=lzma.FORMAT_RAW, filters=filt)
    props = bytes([(pb * 5 + lp) * 9 + lc]) + struct.pack("<I", dict_size)
    return bytes([0x23]) + b"\x03\x01\x01" + number(5) + props, packed


def lzma2_folder(data, dict_byte=10):
    """method 21, props = one dictionary-size byte (reader2.go:296-298)"""
    ds = (2 | (dict_byte & 1)) << (dict_byte // 2 + 11)
    filt = [{"id": lzma.FILTER_LZMA2, "dict_size": ds, "preset": 1}]
    packed = lzma.compress(data, format=lzma.FORMAT_RAW, filters=filt)
    return bytes([0x21]) + b"\x21" + number(1) + bytes([dict_byte]), packed


def copy_folder(data):
    return bytes([0x01]) + b"\x00", bytes(data)


def bcj_lzma_folder(data):
    """two coders (x86 BCJ + LZMA): a chain the front-end must refuse, not misdecode"""
    coder1, packed = lzma_folder(data)
    rec = number(2) + bytes([0x04]) + b"\x03\x03\x01\x03" + coder1 + number(1) + number(0)  # bind pair in 1 <- out 0
    return ("raw", rec), packed


def streams_info(folders, with_substreams=True, folder_crc=False):
    """folders: list of (coder record | ("raw", folder record), packed bytes, [file bytes, ...])"""
    out = bytes([K_PACK_INFO]) + number(0) + number(len(folders)) + bytes([K_SIZE])
    out += b"".join(number(len(p)) for _, p, _ in folders) + bytes([K_END])
    out += bytes([K_UNPACK_INFO, K_FOLDER]) + number(len(folders)) + b"\x00"
    for rec, _, _ in folders:
        out += rec[1] if isinstance(rec, tuple) else number(1) + rec
    out += bytes([K_CODERS_UNPACK_SIZE])
    for rec, _, files in folders:
        total = sum(len(f) for f in files)
        out += (number(total) * 2) if isinstance(rec, tuple) else number(total)
    if folder_crc:
        out += bytes([K_CRC, 1]) + b"".join(struct.pack("<I", zlib.crc32(b"".join(f))) for _, _, f in folders)
    out += bytes([K_END])
    if with_substreams:
        out += bytes([K_SUBSTREAMS, K_NUM_UNPACK_STREAM]) + b"".join(number(len(f)) for _, _, f in folders)
        sizes = b"".join(number(len(x)) for _, _, f in folders for x in f[:-1])
        if sizes:
            out += bytes([K_SIZE]) + sizes
        need = [x for _, _, f in folders if not (len(f) == 1 and folder_crc) for x in f]
        if need:
            out += bytes([K_CRC, 1]) + b"".join(struct.pack("<I", zlib.crc32(x)) for x in need)
        out += bytes([K_END])
    return out + bytes([K_END])


def archive(folders, encoded_header=False, with_substreams=True, folder_crc=False, junk_files_info=True):
    """-> bytes of a .7z file.  encoded_header: the header is LZMA-compressed (7-Zip's default)."""
    packed = b"".join(p for _, p, _ in folders)
    header = bytes([K_HEADER, K_MAIN_STREAMS]) + streams_info(folders, with_substreams, folder_crc)
    if junk_files_info:  # a FilesInfo section the front-end has to ignore: number of files, one dummy property
        nfiles = sum(len(f) for _, _, f in folders)
        header += bytes([K_FILES]) + number(nfiles) + bytes([0x19]) + number(3) + b"\0\0\0" + bytes([K_END])
    header += bytes([K_END])
    if encoded_header:
        rec, hpacked = lzma_folder(header, dict_size=1 << 16)
        hfolder = [(rec, hpacked, [header])]
        enc = bytes([K_ENCODED_HEADER]) + bytes([K_PACK_INFO]) + number(len(packed)) + number(1) + bytes([K_SIZE]) + \
            number(len(hpacked)) + bytes([K_END])
        enc += bytes([K_UNPACK_INFO, K_FOLDER]) + number(1) + b"\x00" + number(1) + rec + bytes([K_CODERS_UNPACK_SIZE]) + \
            number(len(header)) + bytes([K_CRC, 1]) + struct.pack("<I", zlib.crc32(header)) + bytes([K_END]) + bytes([K_END])
        body = packed + hpacked
        next_header = enc
    else:
        body = packed
        next_header = header
    start = struct.pack("<QQI", len(body), len(next_header), zlib.crc32(next_header))
    sig = b"7z\xbc\xaf\x27\x1c" + bytes([0, 4]) + struct.pack("<I", zlib.crc32(start)) + start
    return sig + body + next_header
