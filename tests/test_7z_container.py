"""The .7z front-end (include/xlz.h: xlz_7z_index / xlz_7z_decode).
Most archives are hand-built by tests/sevenzip_craft.py from 7-Zip's published format description; the
packed streams come from liblzma, so the expected output of every folder is known.  Since round 5 also
archives by an independent writer: libarchive's 7zip writer, driven by `cmake -E tar` (bottom of the file).  CPU: the index.  GPU: whole archives as one batch, CRC verification,
encoded headers, and the same folders through the reference's sevenzip constructors."""
import struct
import zlib

import pytest

import corpus
import lzma_amd
from lzma_amd import LzmaError
from sevenzip_craft import archive, bcj_lzma_folder, copy_folder, lzma2_folder, lzma_folder, number


def _folders():
    files = [corpus.plain("T", 40, 70_000), corpus.plain("R", 41, 9_000), corpus.plain("Z", 42, 150_000), b"",
             corpus.plain("M", 43, 200_000), b"tiny", corpus.plain("T", 44, 33_333)]
    solid = files[0:4]                      # one solid LZMA folder with four files (one of them empty)
    r1, p1 = lzma_folder(b"".join(solid), dict_size=1 << 20)
    r2, p2 = lzma2_folder(files[4], dict_byte=12)
    r3, p3 = copy_folder(files[5])
    r4, p4 = lzma_folder(files[6], dict_size=4096, lc=0, lp=2, pb=0)
    return [(r1, p1, solid), (r2, p2, [files[4]]), (r3, p3, [files[5]]), (r4, p4, [files[6]])], b"".join(files)


def test_number_encoding_round_trips_through_the_index():
    for v in (0, 1, 127, 128, 16383, 16384, 1 << 21, (1 << 28) - 1, 1 << 35, (1 << 56) - 1, 1 << 56, (1 << 64) - 1):
        assert len(number(v)) <= 9
    fo, want = _folders()
    a = archive(fo)
    folders, subs, total = lzma_amd.sevenzip_index(a)
    assert total == len(want) and len(folders) == 4
    assert [f["method"] for f in folders] == [1, 2, 3, 1]
    assert [f["n_substreams"] for f in folders] == [4, 1, 1, 1]
    assert folders[0]["dict_size"] == 1 << 20 and folders[0]["props"] == 0x5D
    assert folders[1]["dict_size"] == lzma_amd.DecodeDictSize2(12)
    assert folders[3]["props"] == corpus.props_byte(0, 2, 0) and folders[3]["dict_size"] == 4096
    off = 32
    uoff = 0
    for f, (_, packed, files) in zip(folders, fo):
        assert (f["pack_off"], f["pack_len"]) == (off, len(packed)) and a[off:off + len(packed)] == packed
        assert (f["unpack_off"], f["unpack_len"]) == (uoff, sum(map(len, files)))
        off += len(packed)
        uoff += f["unpack_len"]
    flat = [x for _, _, f in fo for x in f]
    assert subs == [(len(x), zlib.crc32(x)) for x in flat]
    # folder CRCs instead of per-file CRCs, and no SubStreamsInfo at all
    folders2, subs2, _ = lzma_amd.sevenzip_index(archive([fo[1], fo[3]], with_substreams=False, folder_crc=True))
    assert [f["has_crc"] for f in folders2] == [1, 1] and folders2[0]["crc"] == zlib.crc32(fo[1][2][0])
    assert subs2 == [(len(fo[1][2][0]), zlib.crc32(fo[1][2][0])), (len(fo[3][2][0]), zlib.crc32(fo[3][2][0]))]


def test_coder_without_an_input_stream_is_refused():
    """a coder with flag 0x10 and n_in = 0 made the folder claim zero packed streams and
    place_folders index past PackInfo -- a 48-byte archive with valid CRCs crashed xlz_7z_index.  The fixture is that
    archive; header counts that the remaining header bytes cannot back are refused before anything is allocated."""
    import os
    a = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "7z_coder_without_input.7z"), "rb").read()
    assert len(a) == 48
    with pytest.raises(LzmaError) as e:
        lzma_amd.sevenzip_index(a)
    assert e.value.status == lzma_amd.ERR_RESULT

    def with_header(hdr):
        start = struct.pack("<QQI", 0, len(hdr), zlib.crc32(hdr))
        return b"7z\xbc\xaf\x27\x1c" + bytes([0, 4]) + struct.pack("<I", zlib.crc32(start)) + start + hdr
    # n_out = 0, and a lone coder whose only input is bound away (no packed stream left): refused as well
    for coder in (bytes([0x11, 0x00]) + number(1) + number(0),):
        hdr = bytes([1, 4, 7, 11]) + number(1) + b"\x00" + number(1) + coder + bytes([12]) + number(5) + bytes([0, 0, 0])
        with pytest.raises(LzmaError):
            lzma_amd.sevenzip_index(with_header(hdr))
    # 2^24 folders / pack sizes / substreams announced by a header of a few bytes: no allocation, an error
    for hdr in (bytes([1, 4, 7, 11]) + number((1 << 24) - 1) + b"\x00",
                bytes([1, 4, 6]) + number(0) + number((1 << 24) - 1) + bytes([9]),
                bytes([1, 4, 7, 11]) + number(1) + b"\x00" + number(1) + bytes([0x01, 0x00, 12]) + number(5) + bytes([0])
                + bytes([8, 13]) + number((1 << 24) - 1) + bytes([0])):
        with pytest.raises(LzmaError):
            lzma_amd.sevenzip_index(with_header(hdr))


def test_empty_archive():
    start = struct.pack("<QQI", 0, 0, 0)
    a = b"7z\xbc\xaf\x27\x1c" + bytes([0, 4]) + struct.pack("<I", zlib.crc32(start)) + start
    folders, subs, total = lzma_amd.sevenzip_index(a)
    assert folders == [] and subs == [] and total == 0


def test_malformed_and_unsupported_archives_are_refused():
    fo, _ = _folders()
    a = archive(fo)
    for bad in (a[:31], a[:-1], b"", b"\0" * 64, a[:40]):
        with pytest.raises(LzmaError):
            lzma_amd.sevenzip_index(bad)
    for k in (3, 9, 13, 21, len(a) - 5):   # magic, start-header CRC, offsets, header body
        flip = bytearray(a)
        flip[k] ^= 0x10
        with pytest.raises(LzmaError):
            lzma_amd.sevenzip_index(bytes(flip))
    # packed sizes that point past the end of the file
    hdr_at = 32 + struct.unpack("<Q", a[12:20])[0]
    big = bytearray(a)
    i = a.index(bytes([9]), hdr_at)            # kSize of PackInfo
    big[i + 1:i + 2] = number(1 << 40)
    body = bytes(big[32:hdr_at])
    nh = bytes(big[hdr_at:])
    start = struct.pack("<QQI", len(body), len(nh), zlib.crc32(nh))
    forged = a[:8] + struct.pack("<I", zlib.crc32(start)) + start + body + nh
    with pytest.raises(LzmaError):
        lzma_amd.sevenzip_index(forged)
    # an encoded header needs the GPU
    with pytest.raises(LzmaError) as e:
        lzma_amd.sevenzip_index(archive(fo, encoded_header=True))
    assert e.value.status == lzma_amd.ERR_DEVICE
    # a BCJ + LZMA chain is listed as unsupported
    rec, packed = bcj_lzma_folder(b"\x90" * 3000)
    folders, _, _ = lzma_amd.sevenzip_index(archive([(rec, packed, [b"\x90" * 3000])]))
    assert folders[0]["method"] == 0


@pytest.mark.gpu
def test_archives_decode_as_one_batch(ctx):
    fo, want = _folders()
    for enc in (False, True):
        a = archive(fo, encoded_header=enc)
        folders, subs, total = lzma_amd.sevenzip_index(a, ctx)
        assert total == len(want) and len(folders) == 4 and len(subs) == 7
        assert lzma_amd.sevenzip_decode(ctx, a) == want
    # many folders: one stream each in ONE launch
    many = []
    plain = []
    for i in range(300):
        p = corpus.plain("TRMZ"[i % 4], 2000 + i, 2000 + 131 * i)
        plain.append(p)
        rec, packed = lzma_folder(p, dict_size=1 << 16) if i % 3 else lzma2_folder(p, dict_byte=8)
        many.append((rec, packed, [p]))
    big = archive(many, encoded_header=True, folder_crc=True)
    assert lzma_amd.sevenzip_decode(ctx, big) == b"".join(plain)
    # the same archive over three contexts (xlz_7z_decode_multi: folders dealt by compressed bytes; the encoded header is
    # decoded on the first), plus one LZMA2 folder of many dictionary-reset units, which is dealt in slices
    ctxs = [ctx, lzma_amd.Context(0), lzma_amd.Context(0)]
    assert lzma_amd.sevenzip_decode_on(ctxs, big) == b"".join(plain)
    segs = [corpus.plain("T", 5000 + k, 40_000) for k in range(24)]
    packed = corpus.lzma2_concat(segs, dict_size=1 << 16)
    rec = lzma2_folder(b"x", dict_byte=8)[0]      # (the coder record of an LZMA2 folder with a 64 KiB dictionary)
    one = archive([(rec, packed, [b"".join(segs)])], folder_crc=True)
    assert lzma_amd.sevenzip_decode_on(ctxs, one) == b"".join(segs) == lzma_amd.sevenzip_decode(ctx, one)
    assert all(c.last_call_stats()["units"] > 0 for c in ctxs)
    for c in ctxs[1:]:
        c.close()
    # a wrong CRC is caught; a damaged packed stream too
    a = bytearray(archive(fo))
    hdr_at = 32 + struct.unpack("<Q", bytes(a[12:20]))[0]
    nh = bytearray(a[hdr_at:])
    nh[-20] ^= 1                              # inside the per-file CRC list
    start = struct.pack("<QQI", hdr_at - 32, len(nh), zlib.crc32(bytes(nh)))
    bad = bytes(a[:8]) + struct.pack("<I", zlib.crc32(start)) + start + bytes(a[32:hdr_at]) + bytes(nh)
    with pytest.raises(LzmaError) as e:
        lzma_amd.sevenzip_decode(ctx, bad)
    assert e.value.status == lzma_amd.ERR_RESULT
    assert lzma_amd.sevenzip_decode(ctx, bad, verify=False) == want
    a[40] ^= 0xFF
    with pytest.raises(LzmaError):
        lzma_amd.sevenzip_decode(ctx, bytes(a))
    rec, packed = bcj_lzma_folder(b"\x90" * 3000)
    with pytest.raises(LzmaError) as e:
        lzma_amd.sevenzip_decode(ctx, archive([(rec, packed, [b"\x90" * 3000])]))
    assert e.value.status == lzma_amd.ERR_UNSUPPORTED


@pytest.mark.gpu
def test_folders_through_the_sevenzip_constructors(ctx):
    """what bodgit/sevenzip does with the same archive: one NewLZMA(2)DecompressorForSevenZip per folder"""
    fo, want = _folders()
    a = archive(fo, encoded_header=True)
    folders, _, _ = lzma_amd.sevenzip_index(a, ctx)
    out = b""
    for f in folders:
        packed = a[f["pack_off"]: f["pack_off"] + f["pack_len"]]
        if f["method"] == 1:
            r, err = lzma_amd.NewLZMADecompressorForSevenZip(ctx, bytes([f["props"]]) + struct.pack("<I", f["dict_size"]),
                                                             f["unpack_len"], [packed])
        elif f["method"] == 2:
            r, err = lzma_amd.NewLZMA2DecompressorForSevenZip(ctx, bytes([f["props"]]), f["unpack_len"], [packed])
        else:
            out += packed
            continue
        assert err is None
        b, e = r.read_all()
        assert e is None
        out += b
        assert r.Close() is None
    assert out == want


@pytest.mark.gpu
def test_container_fuzz_with_decode(ctx):
    """tools/fuzz_containers.py for a few seconds: mutated .xz files against liblzma (stream by stream), mutated
    .7z archives against their plaintext whenever a verified decode succeeds"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_containers", os.path.join(root, "tools", "fuzz_containers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, n_ok, _ = mod.fuzz(ctx, 6.0, 20251006, verbose=False)
    assert n > 200 and n_ok > 10


# ---- archives by an INDEPENDENT writer: libarchive's 7zip writer through `cmake -E tar` --------------------
def _golden_libarchive():
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return open(os.path.join(g, "libarchive_solid.7z"), "rb").read(), json.load(open(os.path.join(g, "libarchive_solid.json")))


def _check_index(folders, subs, total, exp):
    assert total == exp["folder"]["unpack_size"] and len(folders) == 1
    f = folders[0]
    props = bytes.fromhex(exp["folder"]["props"])
    assert f["method"] == 1 and f["props"] == props[0] and f["dict_size"] == struct.unpack("<I", props[1:5])[0]
    assert (f["pack_off"], f["pack_len"], f["unpack_len"]) == (exp["folder"]["pack_off"], exp["folder"]["pack_len"], total)
    assert f["n_substreams"] == len(exp["substreams"]) and subs == [tuple(x) for x in exp["substreams"]]


def test_an_archive_written_by_libarchive_parses_like_its_own_index():
    """tests/golden/libarchive_solid.7z was written by cmake's bundled libarchive (tests/golden/make_libarchive_7z.py): a
    solid LZMA1 folder (coder 03 01 01, props 5d 00 00 80 00 = lc3 lp0 pb2, 8 MiB dictionary) of four files + an empty one,
    an LZMA-ENCODED header, per-file CRCs.  CPU: the independent plain-Python reader (tests/sevenzip_read.py) and liblzma
    reproduce the committed index and content; the PRODUCT's parser reads the same index from the archive with its header
    stored plainly (an encoded header is a stream like any other: it needs the device -- the GPU test does that)."""
    import hashlib
    import sevenzip_read
    a, exp = _golden_libarchive()
    assert len(a) == exp["archive_bytes"]
    r = sevenzip_read.read(a)
    assert r["encoded"] == exp["encoded_header"]
    m = r["main"]
    f = m["folders"][0]
    assert f["method"].hex() == exp["folder"]["method"] and f["props"].hex() == exp["folder"]["props"]
    assert [list(x) for x in m["substreams"][0]] == exp["substreams"]
    packed = a[32 + m["pack_pos"]: 32 + m["pack_pos"] + m["pack_sizes"][0]]
    content = sevenzip_read.lzma1_decode(packed, f["props"], f["unpack_size"])
    assert hashlib.sha256(content).hexdigest() == exp["sha256"]
    at = 0
    for size, crc in exp["substreams"]:
        assert zlib.crc32(content[at:at + size]) == crc
        at += size
    _check_index(*lzma_amd.sevenzip_index(sevenzip_read.with_plain_header(a)), exp)
    with pytest.raises(LzmaError):      # the encoded header itself: no device, no decode -- loudly
        import torch
        if torch.cuda.is_available():
            raise LzmaError(lzma_amd.ERR_DEVICE, "(a GPU is present: the GPU test covers this)")
        lzma_amd.sevenzip_index(a)


def _cmake_7z(files, tmp_path, name):
    """[(name, bytes)] -> the bytes of a .7z written by cmake / libarchive (entries in the order given)"""
    import shutil
    import subprocess
    if not shutil.which("cmake"):
        pytest.skip("no cmake on this box: nothing here writes a .7z archive")
    d = tmp_path / name
    d.mkdir()
    for n, b in files:
        (d / n).write_bytes(b)
    out = tmp_path / (name + ".7z")
    subprocess.check_call(["cmake", "-E", "tar", "cf", str(out), "--format=7zip"] + [n for n, _ in files], cwd=str(d))
    return out.read_bytes()


@pytest.mark.gpu
def test_archives_written_by_libarchive_decode(ctx, tmp_path):
    """the .7z front-end on archives the test suite did NOT write itself.  The committed fixture and three
    archives written here by cmake's libarchive from seeded files -- one file; 300 small files in one solid folder; 20 MiB
    of far-reaching matches so that the folder's 8 MiB dictionary wraps -- through xlz_7z_decode (CRCs verified, the encoded
    header decoded on the device) and through the reference's plugin entry, NewLZMADecompressorForSevenZip(props,
    unpackSize, readers) (reader1.go:28-61), folder by folder as bodgit/sevenzip calls it."""
    import hashlib
    a, exp = _golden_libarchive()
    _check_index(*lzma_amd.sevenzip_index(a, ctx), exp)
    assert hashlib.sha256(lzma_amd.sevenzip_decode(ctx, a, verify=True)).hexdigest() == exp["sha256"]
    cases = {
        "one": [("only.txt", corpus.plain("T", 8101, 400_000))],
        "many": [("f%03d.%s" % (i, "txt" if i % 3 else "bin"), corpus.plain("TRMZ"[i % 4], 8200 + i, 50 + 97 * i)) for i in range(300)],
        "wrap": [("far.bin", corpus.plain_far(8301, 12 << 20)), ("text.txt", corpus.plain("T", 8302, 8 << 20))],
    }
    for name, files in cases.items():
        arch = _cmake_7z(files, tmp_path, name)
        want = b"".join(b for _, b in files)
        folders, subs, total = lzma_amd.sevenzip_index(arch, ctx)
        assert total == len(want), name
        assert subs == [(len(b), zlib.crc32(b)) for _, b in files if b], name
        assert lzma_amd.sevenzip_decode(ctx, arch, verify=True) == want, name
        out = b""
        for f in folders:   # the reference's own plugin surface
            assert f["method"] == 1, (name, f)
            packed = arch[f["pack_off"]: f["pack_off"] + f["pack_len"]]
            r, err = lzma_amd.NewLZMADecompressorForSevenZip(ctx, bytes([f["props"]]) + struct.pack("<I", f["dict_size"]),
                                                             f["unpack_len"], [packed])
            assert err is None
            b, e = r.read_all()
            assert e is None and r.Close() is None
            out += b
        assert out == want, name
        if name == "wrap":
            assert folders[0]["dict_size"] == 8 << 20 and folders[0]["unpack_len"] > 2 * folders[0]["dict_size"]
    # a damaged folder is caught by the CRCs libarchive wrote
    bad = bytearray(a)
    bad[exp["folder"]["pack_off"] + exp["folder"]["pack_len"] // 2] ^= 0x20
    with pytest.raises(LzmaError):
        lzma_amd.sevenzip_decode(ctx, bytes(bad), verify=True)
