"""The .xz front-end (include/xlz.h: xlz_xz_index / xlz_xz_decode).

The reference has no container code, so the checker here is liblzma (Python's lzma module, the
format's own implementation): every block the index reports must be a raw LZMA2 stream that
liblzma decodes to the right slice of the plaintext, and on the GPU the whole file must decode
to what liblzma gives.
"""
import lzma
import shutil
import subprocess

import pytest

import corpus
import lzma_amd
from lzma_amd import LzmaError


def _three_streams():
    p = corpus.plain("T", 3, 600_000) + corpus.plain("R", 4, 100_000) + corpus.plain("Z", 5, 300_000)
    a = lzma.compress(p[:250_000], format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64)
    b = lzma.compress(p[250_000:650_000], format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32, preset=1)
    c = lzma.compress(p[650_000:], format=lzma.FORMAT_XZ, check=lzma.CHECK_SHA256,
                      filters=[{"id": lzma.FILTER_LZMA2, "preset": 6, "dict_size": 1 << 16, "lc": 2, "lp": 1, "pb": 1}])
    return a + b + b"\0" * 8 + c + b"\0" * 4, p


def _multi_block(block_size=131072):
    if not shutil.which("xz"):
        pytest.skip("xz command not available")
    p = corpus.plain("M", 8, 1_000_000)
    f = subprocess.run(["xz", "-c", "-T2", "--block-size=%d" % block_size], input=p, capture_output=True, check=True).stdout
    return f, p


def _check_blocks_with_liblzma(f, p):
    blocks, total = lzma_amd.xz_index(f)
    assert total == len(p)
    for b in blocks:
        raw = f[b["comp_off"]: b["comp_off"] + b["comp_len"]]
        filt = [{"id": lzma.FILTER_LZMA2, "dict_size": b["dict_size"]}]
        got = lzma.decompress(raw, format=lzma.FORMAT_RAW, filters=filt)
        assert got == p[b["uncomp_off"]: b["uncomp_off"] + b["uncomp_len"]]
    return blocks


def test_index_of_concatenated_streams_with_padding():
    f, p = _three_streams()
    if shutil.which("xz"):  # (Python's lzma.decompress stops at stream padding; the xz tool does not)
        assert subprocess.run(["xz", "-dc"], input=f, capture_output=True, check=True).stdout == p
    blocks = _check_blocks_with_liblzma(f, p)
    assert [b["check_type"] for b in blocks] == [4, 1, 10]
    assert blocks[2]["dict_size"] == 1 << 16


def test_index_of_a_multi_block_file():
    f, p = _multi_block()
    blocks = _check_blocks_with_liblzma(f, p)
    assert len(blocks) == 8 and all(b["uncomp_len"] <= 131072 for b in blocks)


def test_malformed_and_unsupported_files_are_refused():
    f, _ = _three_streams()
    for bad in (f[:-1], f[: len(f) // 2], f[1:], b"", b"\0" * 64):
        with pytest.raises(LzmaError):
            lzma_amd.xz_index(bad)
    flip = bytearray(f)
    flip[-20] ^= 1  # inside the last stream's index / footer
    with pytest.raises(LzmaError):
        lzma_amd.xz_index(bytes(flip))
    bcj = lzma.compress(b"\x90" * 5000, format=lzma.FORMAT_XZ,
                        filters=[{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_LZMA2, "preset": 1}])
    with pytest.raises(LzmaError) as e:
        lzma_amd.xz_index(bcj)
    assert e.value.status == lzma_amd.ERR_UNSUPPORTED


def _vli(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_index_whose_record_sizes_wrap_64_bits_is_refused():
    """four records of 2^62 wrap the sum of the padded sizes to 0 mod 2^64, so a
    crafted index can pass the header-position check and send the block walk out of bounds."""
    import struct
    import zlib
    good = lzma.compress(corpus.plain("T", 9, 5000), format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32)
    blocks, _ = lzma_amd.xz_index(good)
    assert len(blocks) == 1
    # the real block's record, read back from the good file's index
    isz = (struct.unpack("<I", good[-8:-4])[0] + 1) * 4
    ix = len(good) - 12 - isz
    unpadded = ix - 12  # one block, already 4-aligned by its padding or not: recompute from the index
    # parse the real (unpadded, uncompressed) pair
    pos = ix + 2
    vals = []
    for _ in range(2):
        v, sh = 0, 0
        while True:
            b = good[pos]
            pos += 1
            v |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        vals.append(v)
    for order in ("huge-first", "huge-last"):
        recs = [(1 << 62, 1)] * 4
        recs = recs + [tuple(vals)] if order == "huge-first" else [tuple(vals)] + recs
        body = b"\x00" + _vli(len(recs)) + b"".join(_vli(a) + _vli(b) for a, b in recs)
        body += b"\0" * (-len(body) % 4)
        index = body + struct.pack("<I", zlib.crc32(body))
        backward = struct.pack("<I", len(index) // 4 - 1)
        flags = good[-4:-2]
        footer = struct.pack("<I", zlib.crc32(backward + flags)) + backward + flags + b"YZ"
        crafted = good[:ix] + index + footer
        with pytest.raises(LzmaError):
            lzma_amd.xz_index(crafted)


def test_every_structural_bit_flip_is_refused_where_liblzma_refuses():
    """single-bit flips in the stream header, the block header, the block padding, the index and the footer
    of one file per check type: wherever liblzma (stream by stream) refuses the file, xlz_xz_index refuses it
    too -- no GPU needed, the parse is host code"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_containers import strict_xz
    p = bytes((i * 7 + i // 13) % 251 for i in range(3001))   # 3001: the LZMA2 payload needs block padding
    for chk in (lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256):
        x = lzma.compress(p, format=lzma.FORMAT_XZ, check=chk, preset=0)
        n = len(x)
        blocks, _ = lzma_amd.xz_index(x)
        b0 = blocks[0]
        pad = range(b0["comp_off"] + b0["comp_len"], b0["check_off"])
        index_len = (int.from_bytes(x[n - 8:n - 4], "little") + 1) * 4
        structural = list(range(0, b0["comp_off"])) + list(pad) + list(range(n - 12 - index_len, n))
        refused = 0
        for i in structural:
            for bit in range(8):
                y = bytearray(x)
                y[i] ^= 1 << bit
                y = bytes(y)
                if strict_xz(y) is None:
                    refused += 1
                    with pytest.raises(LzmaError):
                        lzma_amd.xz_index(y)
        assert refused > 8 * (len(structural) - 8)
    assert len(pad) > 0


@pytest.mark.gpu
def test_whole_files_decode_as_one_batch(ctx):
    for f, p in (_three_streams(), _multi_block(), _multi_block(65536)):
        assert lzma_amd.xz_decode(ctx, f) == p
    # an empty file body: a stream with no blocks
    e = lzma.compress(b"", format=lzma.FORMAT_XZ)
    assert lzma_amd.xz_decode(ctx, e) == b""


@pytest.mark.gpu
def test_block_check_is_verified(ctx):
    f, p = _multi_block()
    blocks, _ = lzma_amd.xz_index(f)
    bad = bytearray(f)
    bad[blocks[3]["check_off"]] ^= 0x40  # the stored CRC64 of block 3
    with pytest.raises(LzmaError):
        lzma_amd.xz_decode(ctx, bytes(bad))
    assert lzma_amd.xz_decode(ctx, bytes(bad), verify=False) == p
    bad = bytearray(f)
    bad[blocks[2]["comp_off"] + 40] ^= 0x10  # payload damage: wrong bytes or a decode error, never silent
    with pytest.raises(LzmaError):
        lzma_amd.xz_decode(ctx, bytes(bad))
