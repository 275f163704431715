"""CPU: pins the oracle against every vector the reference's own tests hold
(reader1_test.go:15-107, reader2_test.go:12-29) and against liblzma on synthetic data."""
import hashlib
import lzma
import struct

import pytest

import corpus
import oracle

RANDOM_MD5 = "b2d18c4275c394a729607ff9fe0caae7"  # reader1_test.go:107
A_TEXT_MD5 = "57a42eb7f425c13fa644f2618a097ab7"   # LZMA-spec sample plaintext


def test_good_files_decode_without_error(golden):
    exp, data = golden
    for name in ["a.lzma", "a_eos.lzma", "a_eos_and_size.lzma", "a_lp1_lc2_pb1.lzma"]:
        out, st, _ = oracle.lzma1_alone(data[name], 4096)
        assert st == oracle.OK, name            # reader1_test.go:26-49: no error
        assert hashlib.md5(out).hexdigest() == A_TEXT_MD5, name
        assert len(out) == 327


def test_bad_files_return_an_error(golden):
    exp, data = golden
    for name in ["bad_corrupted.lzma", "bad_eos_incorrect_size.lzma", "bad_incorrect_size.lzma"]:
        out, st, _ = oracle.lzma1_alone(data[name], 4096)
        assert st < 0, name                     # reader1_test.go:50-67: some error
        assert st == exp[name]["status"]
        assert len(out) == exp[name]["out_len"]


def test_randomfile_lzma_md5(golden):
    _, data = golden
    out, st, ic = oracle.lzma1_alone(data["randomfile.dat.lzma"], 2 << 20)
    assert st == oracle.OK and len(out) == 1 << 20 and ic == len(data["randomfile.dat.lzma"])
    assert hashlib.md5(out).hexdigest() == RANDOM_MD5  # reader1_test.go:85-105


def test_randomfile_lzma2_md5(golden):
    _, data = golden
    out, st, ic = oracle.lzma2_raw(data["randomfile.dat.lzma2"], 0, 2 << 20)  # NewReader2(r, 0)
    assert st == oracle.OK and len(out) == 1 << 20
    assert hashlib.md5(out).hexdigest() == RANDOM_MD5  # reader2_test.go:12-29


def test_expected_json_matches_oracle(golden):
    exp, data = golden
    for name, e in exp.items():
        if e["format"] == "alone":
            out, st, ic = oracle.lzma1_alone(data[name], 2 << 20)
        else:
            out, st, ic = oracle.lzma2_raw(data[name], e["dict_size"], 2 << 20)
        assert (st, len(out), ic, hashlib.sha256(out).hexdigest()) == \
            (e["status"], e["out_len"], e["in_consumed"], e["sha256"]), name


@pytest.mark.parametrize("family", ["T", "R", "M", "Z"])
@pytest.mark.parametrize("params", [(3, 0, 2, 65536), (2, 1, 1, 1 << 20), (1, 1, 1, 65536), (0, 2, 0, 4096),
                                    (4, 0, 4, 8192)])
def test_oracle_equals_plaintext_and_liblzma(family, params):
    lc, lp, pb, dict_size = params
    p = corpus.plain(family, 7 + lc, 200_000)
    c = corpus.compress_alone(p, dict_size=dict_size, lc=lc, lp=lp, pb=pb, preset=6)
    out, st, ic = oracle.lzma1_alone(c, len(p) + 16)
    assert st == oracle.OK and out == p and ic == len(c)
    assert lzma.decompress(c, format=lzma.FORMAT_ALONE) == p
    # size in the header as well as the end marker (a_eos_and_size flavour)
    c2 = corpus.compress_alone(p, dict_size=dict_size, lc=lc, lp=lp, pb=pb, preset=6, known_size=True)
    out, st, _ = oracle.lzma1_alone(c2, len(p))
    assert st == oracle.OK and out == p
    # raw payload, header fields out of band (sevenzip constructor)
    props, ds, raw = corpus.compress_raw_lzma1(p, dict_size=dict_size, lc=lc, lp=lp, pb=pb)
    out, st, _ = oracle.lzma1_raw(props, ds, 0xFFFFFFFFFFFFFFFF, raw, len(p) + 16)
    assert st == oracle.OK and out == p
    out, st, _ = oracle.lzma1_raw(props, ds, len(p), raw, len(p))
    assert st == oracle.OK and out == p


def test_known_size_no_end_marker():
    p = corpus.plain("T", 3, 100_000)
    c = corpus.alone_known_size_no_eos(p)
    assert c is not None
    out, st, ic = oracle.lzma1_alone(c, len(p))
    assert st == oracle.OK and out == p and ic == len(c)
    assert lzma.decompress(c, format=lzma.FORMAT_ALONE) == p


def test_dict_wraparound():
    # plaintext 40x the dictionary: window.pos wraps many times (window.go:38-41)
    p = corpus.plain("M", 11, 40 * 4096 + 123)
    c = corpus.compress_alone(p, dict_size=4096)
    out, st, _ = oracle.lzma1_alone(c, len(p) + 1)
    assert st == oracle.OK and out == p


def test_lzma2_compressed_chunks_and_resets():
    segs = [corpus.plain("T", 1, 300_000), corpus.plain("R", 2, 70_000), corpus.plain("Z", 3, 500_000),
            corpus.plain("T", 4, 5)]
    c = corpus.lzma2_concat(segs, dict_size=1 << 20)
    want = b"".join(segs)
    out, st, ic = oracle.lzma2_raw(c, 1 << 20, len(want) + 1)
    assert st == oracle.OK and out == want and ic == len(c)
    f = [{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 20}]
    assert lzma.decompress(c, format=lzma.FORMAT_RAW, filters=f) == want
    # multi-chunk single segment: state carried across chunks (no reset / state reset chunks)
    big = corpus.plain("T", 9, 3_000_000)
    c = corpus.compress_raw_lzma2(big, dict_size=1 << 16)
    out, st, _ = oracle.lzma2_raw(c, 1 << 16, len(big))
    assert st == oracle.OK and out == big


def test_header_errors():
    assert oracle.lzma1_alone(b"", 16)[1] == oracle.ERR_HEADER_EOF          # reader1.go:78-81
    assert oracle.lzma1_alone(bytes([225]) + b"\0" * 20, 16)[1] == oracle.ERR_PROPS  # :210-213
    assert oracle.lzma1_alone(bytes([0x5D, 0, 0]), 16)[1] == oracle.ERR_HEADER_EOF
    hdr = bytes([0x5D]) + struct.pack("<I", 65536) + struct.pack("<Q", 10)
    assert oracle.lzma1_alone(hdr, 16)[1] == oracle.ERR_HEADER_EOF          # rangeDec.Init: EOF
    assert oracle.lzma1_alone(hdr + b"\x01\0\0\0\0", 16)[1] == oracle.ERR_RC_INIT  # range_decoder.go:32
    assert oracle.lzma1_alone(hdr + b"\0\0\0", 16)[1] == oracle.ERR_HEADER_EOF


def test_truncated_stream_is_a_clean_eof():
    # SURVEY parity note 4: ReadByte's io.EOF ends the stream without an error
    p = corpus.plain("T", 5, 50_000)
    c = corpus.compress_alone(p)
    out, st, ic = oracle.lzma1_alone(c[: len(c) // 2], len(p))
    assert st == oracle.OK_INPUT_EOF and ic == len(c) // 2
    assert 0 < len(out) < len(p) and p.startswith(out)


def test_out_cap_too_small():
    p = corpus.plain("T", 6, 10_000)
    c = corpus.compress_alone(p)
    out, st, _ = oracle.lzma1_alone(c, 1000)
    assert st == oracle.ERR_OUT_CAP and out == p[:1000]


def test_lzma2_framing_quirks():
    # reader2.go:185-198: control bytes 0x03..0x7F end the stream silently
    assert oracle.lzma2_raw(b"\x03garbage", 0, 16)[1] == oracle.OK
    assert oracle.lzma2_raw(b"", 0, 16)[1] == oracle.ERR_UNEXPECTED_EOF      # :104-110
    assert oracle.lzma2_raw(b"\x01\x00", 0, 16)[1] == oracle.ERR_UNEXPECTED_EOF  # :121-128
    out, st, _ = oracle.lzma2_raw(b"\x01\x00\x02abc", 0, 16)                 # stored chunk, no end byte
    assert out == b"abc" and st == oracle.ERR_UNEXPECTED_EOF
    out, st, _ = oracle.lzma2_raw(b"\x01\x00\x02abc\x00", 0, 16)
    assert out == b"abc" and st == oracle.OK
    out, st, _ = oracle.lzma2_raw(b"\x01\x00\x04ab", 0, 16)                  # stored data cut short
    assert out == b"ab" and st == oracle.ERR_UNEXPECTED_EOF
    # first LZMA chunk without props uses header[5] == 0 -> lc=lp=pb=0 (reader2.go:146-153)
    # (the chunk's 5 bytes only cover rc init, so its first literal hits the chunk limit -> next chunk)
    out, st, ic = oracle.lzma2_raw(b"\x80\x00\x00\x00\x04\x00\x00\x00\x00\x00\x00", 0, 16)
    assert (out, st, ic) == (b"", oracle.OK, 11)
    assert oracle.decode_dict_size2(0) == 4096 and oracle.decode_dict_size2(24) == 16 << 20  # :296-298


def test_mt_batch_driver():
    ps = [corpus.plain("T", i, 20_000) for i in range(6)]
    cs = [corpus.compress_alone(p) for p in ps]
    outs, sts = oracle.decode_batch_mt(cs, [len(p) for p in ps], 3)
    assert sts == [0] * 6 and outs == ps
