"""GPU: the HIP path, called through the C ABI, against the oracle, the committed golden
fixtures and the plaintext (bit-exact: the path is integer/byte work)."""
import hashlib
import struct

import pytest

import corpus
import lzma_amd
import oracle
from lzma_amd import FMT_LZMA_ALONE, FMT_LZMA_RAW, Stream

pytestmark = pytest.mark.gpu

RANDOM_MD5 = "b2d18c4275c394a729607ff9fe0caae7"  # reader1_test.go:107


def _check_against_oracle(ctx, blobs, caps):
    """Every stream: same (bytes, status, in_consumed) as the oracle."""
    got = lzma_amd.decode_batch(ctx, [Stream(b, FMT_LZMA_ALONE, out_cap=c) for b, c in zip(blobs, caps)])
    for i, (b, c) in enumerate(zip(blobs, caps)):
        want = oracle.lzma1_alone(b, c)
        assert got[i][1] == want[1], "status of stream %d: gpu %d oracle %d" % (i, got[i][1], want[1])
        assert got[i][0] == want[0], "bytes of stream %d differ" % i
        assert got[i][2] == want[2], "in_consumed of stream %d" % i
    return got


def test_reference_assets(ctx, golden):
    exp, data = golden
    names = [n for n in exp if exp[n]["format"] == "alone"]
    got = lzma_amd.decode_batch(ctx, [Stream(data[n], FMT_LZMA_ALONE, out_cap=2 << 20) for n in names])
    for n, (out, st, ic) in zip(names, got):
        e = exp[n]
        assert (st, len(out), ic) == (e["status"], e["out_len"], e["in_consumed"]), n
        assert hashlib.sha256(out).hexdigest() == e["sha256"], n
    out = got[names.index("randomfile.dat.lzma")][0]
    assert hashlib.md5(out).hexdigest() == RANDOM_MD5


@pytest.mark.parametrize("family", ["T", "R", "M", "Z"])
def test_cfg2_shape_small_batch(ctx, family):
    # BASELINE config 2 parameters (lc3 lp0 pb2, 64 KiB dict) at an oracle-friendly size
    ps = [corpus.plain(family, 100 + i, 150_000 + 997 * i) for i in range(24)]
    cs = [corpus.compress_alone(p, dict_size=65536, known_size=(i % 2 == 0)) for i, p in enumerate(ps)]
    got = _check_against_oracle(ctx, cs, [len(p) for p in ps])
    for g, p in zip(got, ps):
        assert g[0] == p and g[1] == lzma_amd.OK


@pytest.mark.parametrize("params", [(2, 1, 1, 8 << 20), (1, 1, 1, 65536), (0, 2, 0, 4096), (4, 0, 4, 8192),
                                    (0, 0, 0, 4096), (3, 1, 2, 1 << 16), (0, 4, 4, 1 << 16), (2, 2, 3, 12345)])
def test_other_lc_lp_pb(ctx, params):
    lc, lp, pb, ds = params
    ps = [corpus.plain(f, 31 + i, 120_000) for i, f in enumerate("TRMZ")]
    cs = [corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb) for p in ps]
    got = _check_against_oracle(ctx, cs, [len(p) + 7 for p in ps])
    for g, p in zip(got, ps):
        assert g[0] == p


def test_dict_wraparound_and_odd_dict_sizes(ctx):
    ps, cs = [], []
    for i, ds in enumerate([4096, 4097, 5000, 8191, 65536]):
        p = corpus.plain("M", 50 + i, 30 * ds + 321)
        c = corpus.compress_alone(p, dict_size=max(ds, 4096))
        c = c[:1] + struct.pack("<I", ds) + c[5:]  # header dict size, possibly not a power of two
        ps.append(p)
        cs.append(c)
    _check_against_oracle(ctx, cs, [len(p) + 1 for p in ps])


def test_known_size_without_end_marker(ctx):
    ps = [corpus.plain("T", 70 + i, 90_000 + i) for i in range(4)]
    cs = [corpus.alone_known_size_no_eos(p) for p in ps]
    assert all(c is not None for c in cs)
    got = _check_against_oracle(ctx, cs, [len(p) for p in ps])
    for g, p in zip(got, ps):
        assert g[0] == p and g[1] == lzma_amd.OK


def test_edge_cases_match_oracle(ctx):
    p = corpus.plain("T", 5, 50_000)
    c = corpus.compress_alone(p)
    hdr = bytes([0x5D]) + struct.pack("<I", 65536) + struct.pack("<Q", 10)
    blobs = [
        b"",                                   # constructor: EOF
        bytes([225]) + b"\0" * 20,             # ErrIncorrectProperties
        bytes([0x5D, 0, 0]),                   # header cut
        hdr,                                   # rangeDec.Init: EOF
        hdr + b"\x01\0\0\0\0",                 # first rc byte != 0
        hdr + b"\0\0\0",                       # rc init cut
        c[: len(c) // 2],                      # truncated: clean EOF (parity note 4)
        c[:20],
        c[:14],
        c,                                     # out_cap too small (below)
        corpus.compress_alone(b""),            # empty plaintext, end marker only
        corpus.compress_alone(b"", known_size=True),
        corpus.compress_alone(b"x"),
        c[:13] + bytes(len(c) - 13),           # all-zero payload
        c[:13] + b"\0" + b"\xff" * 200,        # garbage payload
        c[:5] + struct.pack("<Q", len(p) - 100) + c[13:],  # size too small: truncated match / error
        c[:5] + struct.pack("<Q", len(p) + 100) + c[13:],  # size too large: marker with bytesLeft>0
    ]
    caps = [len(p)] * len(blobs)
    caps[9] = 1000
    _check_against_oracle(ctx, blobs, caps)


def test_corrupted_streams_match_oracle(ctx):
    import random
    rnd = random.Random(42)
    blobs = []
    for i in range(48):
        p = corpus.plain("TMZ"[i % 3], 200 + i, 40_000)
        c = bytearray(corpus.compress_alone(p, known_size=(i % 4 == 0)))
        for _ in range(rnd.randint(1, 3)):
            k = rnd.randrange(13, len(c))
            c[k] ^= 1 << rnd.randrange(8)
        blobs.append(bytes(c))
    _check_against_oracle(ctx, blobs, [41_000] * len(blobs))


def test_sevenzip_style_raw_streams(ctx):
    ps = [corpus.plain("T", 90 + i, 77_000) for i in range(3)]
    streams, wants = [], []
    for i, p in enumerate(ps):
        props, ds, raw = corpus.compress_raw_lzma1(p, dict_size=1 << 20)
        size = len(p) if i != 1 else lzma_amd.UNKNOWN_SIZE
        streams.append(Stream(raw, FMT_LZMA_RAW, out_cap=len(p), dict_size=ds, unpack_size=size, props=props))
        wants.append(oracle.lzma1_raw(props, ds, size, raw, len(p)))
    got = lzma_amd.decode_batch(ctx, streams)
    for g, w, p in zip(got, wants, ps):
        assert g == w and g[0] == p


def test_many_streams_round_robin_queue(ctx):
    # more units than resident waves (2560): exercises the persistent grid's work queue
    base = [corpus.compress_alone(corpus.plain("TRMZ"[i % 4], 300 + i, 3000 + 17 * i)) for i in range(64)]
    outs = [oracle.lzma1_alone(b, 8192)[0] for b in base]
    n = 6000
    got = lzma_amd.decode_batch(ctx, [Stream(base[i % 64], FMT_LZMA_ALONE, out_cap=8192) for i in range(n)])
    for i in range(n):
        assert got[i][1] == 0 and got[i][0] == outs[i % 64], i


def test_device_resident_batch_rerun_is_idempotent(ctx):
    ps = [corpus.plain("T", 400 + i, 200_000) for i in range(16)]
    cs = [corpus.compress_alone(p) for p in ps]
    b = lzma_amd.Batch(ctx, [Stream(c, FMT_LZMA_ALONE, out_cap=len(p)) for c, p in zip(cs, ps)])
    for _ in range(3):
        b.run()
    res = b.results()
    assert b.kernel_ms() > 0
    for i, p in enumerate(ps):
        assert res[i][0] == len(p) and res[i][1] == 0
        assert b.download(i, len(p)) == p
    cin, cout, units = b.stats()
    assert cout == sum(len(p) for p in ps) and units == 16
    assert cin == sum(len(c) - 13 for c in cs)
    b.close()
