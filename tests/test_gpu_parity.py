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


# ---------------------------------------------------------------- LZMA2 (reader2.go) ----
from lzma_amd import FMT_LZMA2_RAW  # noqa: E402


def _check_lzma2(ctx, blobs, dicts, caps):
    got = lzma_amd.decode_batch(ctx, [Stream(b, FMT_LZMA2_RAW, out_cap=c, dict_size=ds)
                                      for b, ds, c in zip(blobs, dicts, caps)])
    for i, (b, ds, c) in enumerate(zip(blobs, dicts, caps)):
        want = oracle.lzma2_raw(b, ds, c)
        assert got[i][1] == want[1], "status of stream %d: gpu %d oracle %d" % (i, got[i][1], want[1])
        assert got[i][0] == want[0], "bytes of stream %d differ" % i
        assert got[i][2] == want[2], "in_consumed of stream %d: %d vs %d" % (i, got[i][2], want[2])
    return got


def test_lzma2_reference_asset(ctx, golden):
    exp, data = golden
    out, st, ic = lzma_amd.decode_batch(
        ctx, [Stream(data["randomfile.dat.lzma2"], FMT_LZMA2_RAW, out_cap=2 << 20, dict_size=0)])[0]
    assert st == 0 and ic == len(data["randomfile.dat.lzma2"])
    assert hashlib.md5(out).hexdigest() == RANDOM_MD5  # reader2_test.go:12-29


def test_lzma2_chunk_parallel_units(ctx):
    # cfg4 shape: one stream = many independently compressed segments (dict reset + new props
    # each) -> many units decoded concurrently; plus stored (incompressible) segments
    segs = []
    for i in range(40):
        fam = "TRZM"[i % 4]
        segs.append(corpus.plain(fam, 500 + i, 20_000 + 3001 * (i % 7)))
    blob = corpus.lzma2_concat(segs, dict_size=1 << 16)
    want = b"".join(segs)
    got = _check_lzma2(ctx, [blob], [1 << 16], [len(want)])
    assert got[0][0] == want and got[0][1] == 0


def test_lzma2_state_carried_across_chunks(ctx):
    # a single 3 MB segment: liblzma emits several chunks without dictionary reset, some with
    # and some without state reset -> one unit, model carried from chunk to chunk
    ps = [corpus.plain("T", 600, 3_000_000), corpus.plain("M", 601, 1_500_000), corpus.plain("Z", 602, 2_500_000)]
    blobs = [corpus.compress_raw_lzma2(p, dict_size=1 << 20) for p in ps]
    got = _check_lzma2(ctx, blobs, [1 << 20] * 3, [len(p) for p in ps])
    for g, p in zip(got, ps):
        assert g[0] == p and g[1] == 0


def test_lzma2_other_props_and_small_dict(ctx):
    ps = [corpus.plain("T", 610 + i, 300_000) for i in range(3)]
    blobs = [corpus.compress_raw_lzma2(ps[0], dict_size=4096, lc=0, lp=2, pb=0),
             corpus.compress_raw_lzma2(ps[1], dict_size=8192, lc=4, lp=0, pb=4),
             corpus.lzma2_concat([ps[2][:100_000], ps[2][100_000:]], dict_size=65536, lc=2, lp=1, pb=1)]
    _check_lzma2(ctx, blobs, [4096, 8192, 65536], [len(p) for p in ps])


def test_lzma2_framing_edge_cases_match_oracle(ctx):
    p = corpus.plain("T", 620, 150_000)
    c = corpus.lzma2_concat([p[:50_000], p[50_000:100_000], p[100_000:]], dict_size=1 << 16)
    blobs = [
        b"", b"\x00", b"\x03garbage", b"\x01\x00", b"\x01\x00\x02abc", b"\x01\x00\x02abc\x00", b"\x01\x00\x04ab",
        b"\x02\x00\x02abc\x00",                       # stored, no dict reset as first chunk
        b"\x80\x00\x00\x00\x04\x00\x00\x00\x00\x00\x00",  # LZMA chunk without props first
        b"\xe0\x00\x00\x00\x04\xe1" + b"\0" * 5,      # bad props byte
        b"\xe0\x00\x00\x00\x04\x5d\x01\0\0\0\0\x00",  # rc first byte != 0
        b"\xe0\x00\x00\x00\x02\x5d\x00\0\0",          # rc init cut by the chunk limit
        c[:-1],                                       # missing end byte -> ErrUnexpectedEOF
        c[: len(c) // 2],                             # cut inside a chunk
        c[: len(c) // 3] + c[len(c) // 3 + 5:],       # bytes dropped: headers no longer line up
        c + b"trailing",                              # bytes after the end marker are ignored
        c,                                            # out_cap too small (below)
    ]
    caps = [200_000] * len(blobs)
    caps[-1] = 70_000
    _check_lzma2(ctx, blobs, [1 << 16] * len(blobs), caps)


def test_lzma2_corrupted_streams_match_oracle(ctx):
    import random
    rnd = random.Random(7)
    blobs = []
    for i in range(40):
        segs = [corpus.plain("TMZR"[(i + k) % 4], 700 + 10 * i + k, 15_000) for k in range(4)]
        c = bytearray(corpus.lzma2_concat(segs, dict_size=1 << 16))
        for _ in range(rnd.randint(1, 3)):
            k = rnd.randrange(0, len(c))
            c[k] ^= 1 << rnd.randrange(8)
        blobs.append(bytes(c))
    _check_lzma2(ctx, blobs, [1 << 16] * len(blobs), [80_000] * len(blobs))


# ------------------------------------------- pull-style readers (reader1.go / reader2.go) ----
def test_reader1_read_semantics(ctx):
    p = corpus.plain("T", 800, 100_000)
    r, err = lzma_amd.NewReader1(ctx, corpus.compress_alone(p))
    assert err is None
    got = []
    while True:                                   # io.Copy with a 32 KiB buffer (reader1_test.go:79)
        b, e = r.Read(32768)
        got.append(b)
        if e is lzma_amd.io_EOF:
            break
        assert e is None
    assert b"".join(got) == p
    b, e = r.Read(10)
    assert b == b"" and e is lzma_amd.io_EOF      # reader1.go:239-243


def test_reader1_reference_assets(ctx, golden):
    exp, data = golden
    for name in ["a.lzma", "a_eos.lzma", "a_eos_and_size.lzma", "a_lp1_lc2_pb1.lzma"]:
        r, err = lzma_amd.NewReader1(ctx, data[name])
        assert err is None                        # checkErr1: NoError (reader1_test.go:26-49)
        out, e = r.read_all()
        assert e is None and len(out) == 327      # checkErr2: NoError
    for name in ["bad_corrupted.lzma", "bad_eos_incorrect_size.lzma", "bad_incorrect_size.lzma"]:
        r, err = lzma_amd.NewReader1(ctx, data[name])
        assert err is None                        # constructor succeeds (reader1_test.go:50-67)
        out, e = r.read_all()
        assert isinstance(e, lzma_amd.LzmaError) and e.status == lzma_amd.ErrResultError


def test_reader_constructor_errors(ctx):
    for blob, status in [(b"", lzma_amd.ERR_HEADER_EOF), (bytes([225]) + b"\0" * 20, lzma_amd.ERR_PROPS),
                         (bytes([0x5D, 0, 0]), lzma_amd.ERR_HEADER_EOF),
                         (bytes([0x5D]) + struct.pack("<IQ", 65536, 10) + b"\x01\0\0\0\0", lzma_amd.ERR_RC_INIT)]:
        r, err = lzma_amd.NewReader1(ctx, blob)
        assert r is None and err.status == status
    r, err = lzma_amd.NewReader2(ctx, b"", 0)
    assert r is None and err.status == lzma_amd.ERR_UNEXPECTED_EOF     # reader2.go:104-110


def test_reader2_and_sevenzip_constructors(ctx, golden):
    exp, data = golden
    r, err = lzma_amd.NewReader2(ctx, data["randomfile.dat.lzma2"], 0)   # reader2_test.go:19
    assert err is None
    out, e = r.read_all()
    assert e is None and hashlib.md5(out).hexdigest() == RANDOM_MD5
    p = corpus.plain("T", 801, 60_000)
    props, ds, raw = corpus.compress_raw_lzma1(p, dict_size=1 << 20)
    rc, err = lzma_amd.NewLZMADecompressorForSevenZip(ctx, bytes([props]) + struct.pack("<I", ds), len(p), [raw])
    assert err is None
    out, e = rc.read_all()
    assert e is None and out == p
    assert rc.Close() is None
    assert rc.Close().status == lzma_amd.ERR_CLOSED                    # readcloser.go:17-19
    assert rc.Read(4)[1].status == lzma_amd.ERR_CLOSED                 # readcloser.go:31-33
    _, err = lzma_amd.NewLZMADecompressorForSevenZip(ctx, bytes([props]) + struct.pack("<I", ds), len(p), [raw, raw])
    assert err.status == lzma_amd.ERR_NEED_ONE_READER                  # reader1.go:33-35
    _, err = lzma_amd.NewLZMA2DecompressorForSevenZip(ctx, b"\x10\x00", 0, [raw])
    assert err.status == lzma_amd.ERR_INSUFFICIENT_PROPS               # reader2.go:54-56
    c2 = corpus.compress_raw_lzma2(p, dict_size=1 << 16)
    rc, err = lzma_amd.NewLZMA2DecompressorForSevenZip(ctx, bytes([10]), 0, [c2])  # dict byte 10 -> 128 KiB
    assert err is None
    out, e = rc.read_all()
    assert e is None and out == p


def test_multi_device_driver_on_gpu(ctx):
    from lzma_amd import multigpu
    ps = [corpus.plain("TRMZ"[i % 4], 900 + i, 30_000) for i in range(12)]
    streams = [Stream(corpus.compress_alone(p), out_cap=len(p)) for p in ps]
    res = multigpu.decode_batch_multi(streams, [0, 0])   # two host threads, two contexts, one GPU
    assert [r[0] for r in res] == ps and all(r[1] == 0 for r in res)
    # the same inside the library: xlz_decode_batch_multi over three contexts (all on this GPU),
    # ragged sizes and one bad stream; results must come back in input order
    ps2 = [corpus.plain("TRMZ"[i % 4], 950 + i, 1000 + 9000 * (i % 7)) for i in range(23)]
    streams2 = [Stream(corpus.compress_alone(p), out_cap=len(p)) for p in ps2]
    streams2[5] = Stream(streams2[5].data[:40], out_cap=len(ps2[5]))
    ctxs = [ctx, lzma_amd.Context(0), lzma_amd.Context(0)]
    res2 = lzma_amd.decode_batch_on(ctxs, streams2)
    for i, (r, p) in enumerate(zip(res2, ps2)):
        if i == 5:
            assert r == oracle.lzma1_alone(streams2[5].data, len(p))
        else:
            assert r[0] == p and r[1] == 0, i
    assert lzma_amd.decode_batch_on([ctx], streams2[:3]) == lzma_amd.decode_batch(ctx, streams2[:3])


def test_units_of_one_stream_and_blocks_of_one_file_over_three_contexts(ctx):
    """SURVEY 8e, second clause: ONE LZMA2 stream of 4096 dictionary-reset units through
    xlz_decode_batch_multi with three contexts (all on this GPU): every context decodes a run of units -- a slice of the
    compressed input into a disjoint slice of the caller's buffer --, bytes / status / in_consumed equal the oracle's,
    every context's last call decoded units.  Then the same stream damaged (a unit that no longer decodes to what its
    headers announce: the call falls back to the whole stream on one context and still equals the oracle), a stream that
    reads behind a dictionary reset into another slice's bytes, and a multi-block .xz file through xlz_xz_decode_multi."""
    import lzma
    from lzma_craft import long_stale_lzma2_stream
    enc = {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2}
    segs = [corpus.plain("TRMZ"[k % 4] if k % 16 == 0 else "T", 7000 + k, 9000 + 700 * (k % 13)) for k in range(4096)]
    blob = corpus.lzma2_concat(segs, dict_size=1 << 16, preset=enc)
    plain = b"".join(segs)
    want = oracle.lzma2_raw(blob, 1 << 16, len(plain))
    assert want == (plain, 0, len(blob))
    ctxs = [ctx, lzma_amd.Context(0), lzma_amd.Context(0)]
    s = Stream(blob, lzma_amd.FMT_LZMA2_RAW, out_cap=len(plain), dict_size=1 << 16)
    items = lzma_amd.multi_plan(3, [s])
    assert len(items) == 12 and {it["context"] for it in items} == {0, 1, 2}
    res = lzma_amd.decode_batch_on(ctxs, [s])
    assert res[0] == want
    units = [c.last_call_stats()["units"] for c in ctxs]
    assert all(u > 0 for u in units) and sum(units) == 4096, units
    # with other streams around it (dealt whole), results in input order
    small = [corpus.plain("T", 7900 + i, 20_000) for i in range(30)]
    mixed = [Stream(corpus.compress_alone(p), out_cap=len(p)) for p in small[:15]] + [s] + \
            [Stream(corpus.compress_alone(p), out_cap=len(p)) for p in small[15:]]
    res = lzma_amd.decode_batch_on(ctxs, mixed)
    assert res[15] == want and [r[0] for r in res[:15] + res[16:]] == small and all(r[1] == 0 for r in res)
    # a damaged unit in the middle: the slices no longer come back as announced -> the whole stream on one context
    for at in (len(blob) // 2, len(blob) // 3 + 17, len(blob) - 5000):
        bad = bytearray(blob)
        bad[at] ^= 0x40
        bad = bytes(bad)
        got = lzma_amd.decode_batch_on(ctxs, [Stream(bad, lzma_amd.FMT_LZMA2_RAW, out_cap=len(plain), dict_size=1 << 16)])[0]
        assert got == oracle.lzma2_raw(bad, 1 << 16, len(plain)), at
    # the stream cut short (no end byte) and with too little room
    for cut, cap in ((len(blob) - 1, len(plain)), (len(blob) * 2 // 3, len(plain)), (len(blob), len(plain) - 100_000)):
        got = lzma_amd.decode_batch_on(ctxs, [Stream(blob[:cut], lzma_amd.FMT_LZMA2_RAW, out_cap=cap, dict_size=1 << 16)])[0]
        assert got == oracle.lzma2_raw(blob[:cut], 1 << 16, cap), (cut, cap)
    # copies that read BEHIND a dictionary reset (window.go:135-140 keeps the buffer): the bytes belong to another slice
    # (behind four incompressible dictionary-reset segments, so that the stream is worth cutting and the reads happen in a
    #  slice that does not begin at the stream's start: no slice can settle them -> the whole stream on one context)
    pre = [corpus.plain("R", 8000 + k, 30_000) for k in range(4)]
    for ds in (4096, 65536):
        sb, sw = long_stale_lzma2_stream(ds, seed=ds + 1)
        assert sb[0] == 1 or sb[0] >= 0xE0
        both = corpus.lzma2_concat(pre, dict_size=ds, preset=enc)[:-1] + sb
        ss = Stream(both, lzma_amd.FMT_LZMA2_RAW, out_cap=120_000 + len(sw) + 100, dict_size=ds)
        assert len(lzma_amd.multi_plan(3, [ss])) > 1
        assert lzma_amd.decode_batch_on(ctxs, [ss])[0] == oracle.lzma2_raw(both, ds, ss.out_cap) == (b"".join(pre) + sw, 0, len(both))
    # an .xz file of 48 blocks: the blocks are the items
    blocks = [corpus.plain("T", 8100 + i, 150_000 + 1000 * i) for i in range(48)]
    filt = [dict(corpus.lzma1_filters(dict_size=1 << 20, preset=enc)[0], id=lzma.FILTER_LZMA2)]
    xz = b"".join(lzma.compress(p, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64, filters=filt) for p in blocks)
    assert lzma_amd.xz_decode_on(ctxs, xz) == b"".join(blocks) == lzma_amd.xz_decode(ctx, xz)
    assert all(c.last_call_stats()["units"] > 0 for c in ctxs)
    for c in ctxs[1:]:
        c.close()


# ------------------------------------------------ models too large for LDS (lc+lp > 6) ----
def test_large_lc_lp_models_live_in_hbm(ctx):
    """The reference accepts lc <= 8, lp <= 4 (reader1.go:210-221): up to 0x300 << 12 probs.
    liblzma refuses to ENCODE lc+lp > 4, so these streams are ordinary payloads relabelled with
    a large-model props byte: the decode is garbage, but deterministic garbage, and it must
    equal the oracle's byte for byte (both walk the same 12-bit literal contexts)."""
    import random
    rnd = random.Random(5)
    blobs = []
    for i, (lc, lp, pb) in enumerate([(8, 4, 2), (7, 0, 0), (8, 0, 4), (4, 4, 1), (3, 4, 2), (6, 1, 0)]):
        p = corpus.plain("TRMZ"[i % 4], 1100 + i, 30_000)
        c = bytearray(corpus.compress_alone(p))
        c[0] = corpus.props_byte(lc, lp, pb)
        if i % 2:
            c[5:13] = struct.pack("<Q", 20_000)   # defined size
        blobs.append(bytes(c))
    # plus pure noise payloads
    for i in range(4):
        hdr = bytes([corpus.props_byte(8, 4, 4)]) + struct.pack("<IQ", 1 << 16, 50_000)
        blobs.append(hdr + b"\x00" + bytes(rnd.randrange(256) for _ in range(4000)))
    # an all-zero payload decodes to a long run of literals in every parameter set (Code stays 0)
    for lc, lp, pb in [(8, 4, 4), (7, 2, 0), (3, 0, 2)]:
        blobs.append(bytes([corpus.props_byte(lc, lp, pb)]) + struct.pack("<IQ", 1 << 16, 0xFFFFFFFFFFFFFFFF) +
                     bytes(3000))
    got = _check_against_oracle(ctx, blobs, [60_000] * len(blobs))
    assert any(len(g[0]) > 10_000 for g in got)
    # mixed batch: LDS-model and HBM-model streams together
    p = corpus.plain("T", 1200, 40_000)
    mixed = [corpus.compress_alone(p), blobs[0], corpus.compress_alone(p, lc=2, lp=2, pb=0), blobs[7]]
    _check_against_oracle(ctx, mixed, [60_000] * 4)


def test_lzma2_many_small_units_share_one_output_range(ctx):
    # hundreds of tiny dictionary-reset units packed back to back in ONE output range: a copy
    # that stored past its unit's end would corrupt the neighbour (64-lane rows)
    segs = [corpus.plain("ZTZM"[i % 4], 1300 + i, 700 + 37 * (i % 23)) for i in range(600)]
    blob = corpus.lzma2_concat(segs, dict_size=1 << 16, preset=1)
    want = b"".join(segs)
    for _ in range(3):
        got = _check_lzma2(ctx, [blob], [1 << 16], [len(want)])
        assert got[0][0] == want


def _stored_units(blob, unit=256 << 10):
    """units of an LZMA2 stream made of stored chunks alone, by scan_lzma2's rule: a cut at every chunk that resets the
    dictionary and at every other chunk once `unit` bytes have gathered since the last cut"""
    pos, units, acc = 0, 1, 0
    while blob[pos] in (1, 2):
        n = ((blob[pos + 1] << 8) | blob[pos + 2]) + 1
        if pos and (blob[pos] == 1 or acc >= unit):
            units, acc = units + 1, 0
        acc += n
        pos += 3 + n
    assert blob[pos] == 0
    return units


def test_lzma2_runs_of_stored_chunks_split_at_their_dictionary_resets(ctx):
    """The shape of the reference's own LZMA2 benchmark file (randomfile.dat.lzma2, reader2_test.go:31-36): stored
    chunks only.  A stored chunk that resets the dictionary may start a unit when no LZMA chunk behind it continues an
    earlier model (scan_lzma2): 60 incompressible segments are 60 units, not one wave copying everything; the same
    segments with a compressed chunk WITHOUT new properties behind a stored reset must not be cut there -- both against
    the oracle, and through a reader."""
    from lzma_craft import Encoder, Window, lzma2_lzma_chunk, lzma2_stored, props_byte
    segs = [corpus.plain("R", 1500 + i, 70_000 + 997 * (i % 7)) for i in range(60)]
    blob = corpus.lzma2_concat(segs, dict_size=1 << 16, preset=0)
    want = b"".join(segs)
    assert blob[0] == 1 and blob.count(b"\x01", 0, 1) == 1          # stored, dictionary reset
    b = lzma_amd.Batch(ctx, [Stream(blob, FMT_LZMA2_RAW, out_cap=len(want), dict_size=1 << 16)])
    b.run()
    res = b.results()
    assert res[0][0] == len(want) and res[0][1] == 0 and res[0][2] == len(blob)
    assert b.download(0, len(want)) == want
    # units: a run of stored chunks that nothing reads is cut at every dictionary reset (and wherever 256 KiB have gathered:
    # never inside these 70 KB segments)
    assert b.stats()[2] == _stored_units(blob) == 60
    b.close()
    # mixed: text segments (0xE0 chunks), stored runs, and a crafted tail whose 0x80 chunk (no new properties) follows
    # a stored reset: that candidate must NOT become a cut (the chunk continues the model of the chunks before)
    segs2 = [corpus.plain("TR"[i % 2], 1600 + i, 50_000) for i in range(9)]
    w = Window(1 << 16)
    e = Encoder(3, 0, 2, 1 << 16, window=w)
    for i in range(500):
        e.literal(97 + i % 7)
    e.match(100, 50)
    tail = lzma2_lzma_chunk(0xE0, len(w.total), e.payload(), props_byte(3, 0, 2))
    n0 = len(w.total)
    w.reset()
    for x in b"stored-after-reset":
        w.put(x)
    tail += lzma2_stored(b"stored-after-reset", dict_reset=True)
    e.new_chunk()
    n1 = len(w.total)
    for i in range(300):
        e.literal(65 + i % 5)
    e.rep(0, 20)
    tail += lzma2_lzma_chunk(0x80, len(w.total) - n1, e.payload()) + b"\x00"
    blob2 = corpus.lzma2_concat(segs2, dict_size=1 << 16, preset=0)[:-1] + tail
    want2 = oracle.lzma2_raw(blob2, 1 << 16, 1 << 20)
    assert want2[1] == 0 and want2[0] == b"".join(segs2) + bytes(w.total)
    got = _check_lzma2(ctx, [blob, blob2], [1 << 16, 1 << 16], [len(want), len(want2[0])])
    assert got[0][0] == want and got[1] == want2
    r, err = lzma_amd.NewReader2(ctx, blob, 1 << 16)
    out, e2 = r.read_all(chunk=100_000)
    assert e2 is None and out == want


def test_stored_chunks_of_every_length_and_alignment(ctx):
    """The stored-chunk copy moves 16 bytes per lane between an ALIGNED destination and a source at any byte offset
    (xlz_kernel.hip: stored_copy -- head bytes, 16-byte groups funnelled from aligned dwords, tail bytes).  Streams made
    of stored chunks of 1 .. 65 536 bytes in shuffled order put every (source mod 4, destination mod 16, length mod 16)
    combination in front of it, inside units (dictionary resets) and across them; plus an output capacity that ends
    inside a chunk and sources cut short inside a chunk (window.ReadFrom delivers what is there, window.go:142-155)."""
    import random
    from lzma_craft import lzma2_stored
    lens = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 18, 19, 31, 32, 33, 47, 48, 49, 63, 64, 65, 66, 100, 127, 128, 129, 255, 256, 257,
            1023, 1024, 1025, 4093, 4096, 4099, 16383, 16385, 65535, 65536]
    blobs, caps = [], []
    for k in range(12):
        rng = random.Random(9000 + k)
        order = lens * 2
        rng.shuffle(order)
        data = corpus.plain("R", 9100 + k, sum(order))
        blob, pos = b"", 0
        for j, n in enumerate(order):
            blob += lzma2_stored(data[pos:pos + n], dict_reset=(j == 0 or rng.random() < 0.15))
            pos += n
        blob += b"\x00"
        blobs.append(blob)
        caps.append(len(data))
        if k % 3 == 1:   # the capacity ends inside a chunk (ERR_OUT_CAP after the bytes that fit)
            blobs.append(blob)
            caps.append(len(data) - rng.randrange(1, 70000))
        if k % 3 == 2:   # the source ends inside a chunk
            blobs.append(blob[: len(blob) - rng.randrange(2, 60000)])
            caps.append(len(data))
    got = _check_lzma2(ctx, blobs, [1 << 16] * len(blobs), caps)
    assert got[0][1] == 0 and len(got[0][0]) == caps[0]
    # the same through pull readers (sessions: the destination is a sliding device window)
    for blob in blobs[:3]:
        want = oracle.lzma2_raw(blob, 1 << 16, 8 << 20)
        r, err = lzma_amd.NewReader2(ctx, blob, 1 << 16)
        out, e2 = r.read_all(chunk=33_333)
        assert e2 is None and out == want[0]


def test_runs_of_stored_chunks_are_cut_at_every_chunk_unless_a_later_chunk_reads_them(ctx):
    """scan_lzma2: inside a run of stored chunks that ends at a dictionary reset or at the end of the stream every chunk
    boundary may start a unit (a stored chunk reads no history, window.go:142-155): ONE incompressible stream -- 0x01 once,
    then 0x02 chunks, what xz writes for a random file -- is copied by a wave per 256 KiB.  An LZMA chunk that keeps the
    dictionary (0xC0: new properties, state reset, NO dictionary reset) reads the stored bytes in front of it through its
    matches: its run must stay in its unit.  All against the oracle; the unit counts say which cut was made."""
    from lzma_craft import Encoder, Window, lzma2_lzma_chunk, lzma2_stored, props_byte
    data = corpus.plain("R", 7700, 1_000_003)
    chunks = [data[i:i + 65536] for i in range(0, len(data), 65536)]
    pure = b"".join(lzma2_stored(c, dict_reset=(j == 0)) for j, c in enumerate(chunks)) + b"\x00"
    b = lzma_amd.Batch(ctx, [Stream(pure, FMT_LZMA2_RAW, out_cap=len(data), dict_size=1 << 16)])
    b.run()
    res = b.results()
    assert res[0] == (len(data), 0, len(pure)) and b.download(0, len(data)) == data
    assert len(chunks) == 16 and b.stats()[2] == _stored_units(pure) == 4   # a cut wherever 256 KiB have gathered
    b.close()
    # stored run, then an LZMA chunk WITHOUT dictionary reset whose matches copy stored bytes from up to 65 000 back
    w = Window(1 << 16)
    head = data[:600_000]
    blob = b""
    for j in range(0, len(head), 65536):
        blob += lzma2_stored(head[j:j + 65536], dict_reset=(j == 0))
    for x in head:
        w.put(x)
    e = Encoder(3, 0, 2, 1 << 16, window=w)
    n0 = len(w.total)
    e.match(65_000, 273)
    e.match(40_000, 100)
    for i in range(40):
        e.literal(48 + i % 10)
    e.match(65_536, 17)
    e.rep(1, 30)
    kept = blob + lzma2_lzma_chunk(0xC0, len(w.total) - n0, e.payload(), props_byte(3, 0, 2)) + b"\x00"
    want = bytes(w.total)
    o = oracle.lzma2_raw(kept, 1 << 16, len(want) + 100)
    assert o[1] == 0 and o[0] == want
    b = lzma_amd.Batch(ctx, [Stream(kept, FMT_LZMA2_RAW, out_cap=len(want), dict_size=1 << 16)])
    b.run()
    assert b.results()[0] == (len(want), 0, len(kept)) and b.download(0, len(want)) == want
    assert b.stats()[2] == 1                       # the run is read by the chunk behind it: one unit
    b.close()
    # the same run in front of a chunk that DOES reset the dictionary (0xE0), and a second pure run behind that chunk
    e2 = Encoder(3, 0, 2, 1 << 16)
    for i in range(300):
        e2.literal(97 + i % 13)
    e2.match(200, 60)
    tail_data = corpus.plain("R", 7701, 150_000)
    cut = blob + lzma2_lzma_chunk(0xE0, 360, e2.payload(), props_byte(3, 0, 2))
    for j in range(0, len(tail_data), 65536):
        cut += lzma2_stored(tail_data[j:j + 65536], dict_reset=(j == 0))
    cut += b"\x00"
    o = oracle.lzma2_raw(cut, 1 << 16, 1 << 20)
    assert o[1] == 0 and o[0][:600_000] == head and o[0][600_360:] == tail_data
    got = _check_lzma2(ctx, [pure, kept, cut], [1 << 16] * 3, [len(data), len(want), len(o[0])])
    assert got[2][0] == o[0]
    b = lzma_amd.Batch(ctx, [Stream(cut, FMT_LZMA2_RAW, out_cap=len(o[0]), dict_size=1 << 16)])
    b.run()
    assert b.results()[0][1] == 0 and b.stats()[2] == 3 + 1 + 1   # ten stored chunks in three units, the LZMA chunk, the run behind it
    b.close()
    # readers: the parallel refill path takes the finer units too
    r, err = lzma_amd.NewReader2(ctx, pure, 1 << 16)
    out, e3 = r.read_all(chunk=77_777)
    assert e3 is None and out == data


# ------------------------------------------- BASELINE-sized batches: size-independent properties ----
def test_baseline_shape_roundtrip_and_idempotence(ctx):
    """4096 streams (the stream count of BASELINE config 2; 64 KiB each so the corpus builds in
    seconds): decode(compress(x)) == x for every stream by SHA-256, a second run of the same
    device-resident batch gives identical bytes (idempotence), and the batch's byte accounting
    adds up (sum of out_len == sum of plaintext sizes, in_consumed == payload sizes)."""
    n, size = 4096, 65536
    comp, digests = corpus.make_alone_batch("M", n, size, base_seed=7000, preset=1)
    b = lzma_amd.Batch(ctx, [Stream(c, FMT_LZMA_ALONE, out_cap=size) for c in comp])
    b.run()
    res = b.results()
    assert all(r[1] == 0 and r[0] == size for r in res)
    first = [hashlib.sha256(b.download(i, size)).digest() for i in range(n)]
    assert first == digests
    b.run()
    assert [hashlib.sha256(b.download(i, size)).digest() for i in range(0, n, 7)] == first[::7]
    cin, cout, units = b.stats()
    assert cout == n * size and units == n and cin == sum(len(c) - 13 for c in comp)
    b.close()


def test_concurrent_readers_are_coalesced_into_batches(xlz_so):
    """many threads each doing the reference's `NewReader1(src)` + io.Copy get
    decoded as a few GPU batches instead of one launch per reader."""
    import threading
    c2 = lzma_amd.Context(0)
    c2.enable_batching(window_us=20_000, max_streams=256)
    ps = [corpus.plain("TRMZ"[i % 4], 1500 + i, 20_000 + 501 * i) for i in range(48)]
    blobs = [corpus.compress_alone(p, known_size=(i % 3 == 0)) for i, p in enumerate(ps)]
    blobs[5] = blobs[5][:200]                       # truncated: clean EOF, short output
    bad = bytearray(blobs[9]); bad[40] ^= 0x55; blobs[9] = bytes(bad)
    outs, errs = [None] * 48, [None] * 48

    def work(i):
        r, err = lzma_amd.NewReader1(c2, blobs[i])
        assert err is None
        outs[i], errs[i] = r.read_all(chunk=4096 + i)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(48)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(48):
        want = oracle.lzma1_alone(blobs[i], len(ps[i]) * 8 + 65536)
        assert outs[i] == want[0], i
        assert (errs[i] is None) == (want[1] >= 0), i
    batches, streams = c2.batching_stats()
    assert streams >= 48 and batches <= 24, (batches, streams)
    c2.close()


def test_host_pipeline_big_and_ragged_streams(ctx):
    """xlz_decode_batch's staged host path (pinned ring, chunked D2H, scatter threads): a stream
    larger than one ring slot (64 MiB), many small ones around it, a truncated and an empty one --
    every byte against the plaintext / the oracle, twice (the pinned pools are reused)."""
    import threading
    big = corpus.plain("Z", 901, 80 * 1024 * 1024 + 12345)
    items = [(corpus.compress_alone(big, dict_size=1 << 20), big)]
    for k in range(40):
        p = corpus.plain("TRMZ"[k % 4], 910 + k, 1000 + 37777 * k)
        items.append((corpus.compress_alone(p), p))
    trunc = items[5][0][: len(items[5][0]) // 2]
    streams = [Stream(c, FMT_LZMA_ALONE, out_cap=len(p)) for c, p in items]
    streams.insert(3, Stream(trunc, FMT_LZMA_ALONE, out_cap=len(items[5][1])))
    streams.insert(7, Stream(b"", FMT_LZMA_ALONE, out_cap=10))
    want = [(p, 0) for _, p in items]
    wt = oracle.lzma1_alone(trunc, len(items[5][1]))
    want.insert(3, (wt[0], wt[1]))
    we = oracle.lzma1_alone(b"", 10)
    want.insert(7, (we[0], we[1]))
    for _ in range(2):
        got = lzma_amd.decode_batch(ctx, streams)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g[1] == w[1], i
            assert hashlib.sha256(g[0]).digest() == hashlib.sha256(w[0]).digest(), i

    # two host threads share the context (one staged call at a time inside)
    errs = []

    def run():
        try:
            r = lzma_amd.decode_batch(ctx, streams[1:12])
            for (g, w) in zip(r, want[1:12]):
                assert g[1] == w[1] and g[0] == w[0]
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run) for _ in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_differential_fuzz_against_the_oracle(ctx):
    """tools/fuzz_gpu.py for a few seconds: random families, sizes around the fast path's margins,
    lc/lp/pb, odd dictionary sizes, caps, truncations, byte flips, damaged LZMA2 framing -- bytes,
    status and consumed input against the oracle.  (A five-minute run of the same tool, 53 760
    streams, is recorded in DESIGN.md.)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(root, "tools", "fuzz_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, n_bad = mod.fuzz(ctx, budget=8.0, seed=20251004, per_round=96, verbose=False)
    assert n >= 96 and n_bad > 0


def test_large_mixed_call_of_decode_batch(ctx):
    """one xlz_decode_batch call with five wave rounds of work: 20 000 streams / 150 MiB of input, LZMA1
    and multi-unit LZMA2 streams, ragged sizes, a truncated, a corrupted, an empty and an
    out-of-room stream, plus a malformed LZMA2 stream that needs the exact re-run -- every result
    equals the plaintext (the oracle for the bad ones), twice (the pinned pools are reused)."""
    from lzma_amd import FMT_LZMA2_RAW
    base = []
    for i in range(80):
        p = corpus.plain("R", 8000 + i, 6000 + 113 * i)                # incompressible: input-heavy
        base.append((Stream(corpus.compress_alone(p, preset=0), FMT_LZMA_ALONE, out_cap=len(p)), p))
    for i in range(30):
        p = corpus.plain("TMZ"[i % 3], 8100 + i, 20_000 + 1000 * i)
        base.append((Stream(corpus.compress_alone(p, preset=0, known_size=(i % 2 == 0)), FMT_LZMA_ALONE, out_cap=len(p)), p))
    for i in range(10):
        segs = [corpus.plain("TRZ"[(i + k) % 3], 8200 + 10 * i + k, 9_000) for k in range(4)]
        base.append((Stream(corpus.lzma2_concat(segs, dict_size=1 << 16, preset=0), FMT_LZMA2_RAW, out_cap=36_000,
                            dict_size=1 << 16), b"".join(segs)))
    n = 20_000
    streams = [base[i % len(base)][0] for i in range(n)]
    want = [(base[i % len(base)][1], 0) for i in range(n)]
    good = base[0][0].data
    bad = bytearray(base[81][0].data)
    bad[len(bad) // 2] ^= 0x20
    specials = {7: Stream(good[: len(good) // 2], FMT_LZMA_ALONE, out_cap=60_000),
                4000: Stream(bytes(bad), FMT_LZMA_ALONE, out_cap=40_000),
                9555: Stream(b"", FMT_LZMA_ALONE, out_cap=10),
                n - 1: Stream(good, FMT_LZMA_ALONE, out_cap=1000)}
    from test_crafted_streams import crafted_lzma2
    name, blob, ds, cap, _ = crafted_lzma2()[2]
    specials[12345] = Stream(blob, FMT_LZMA2_RAW, out_cap=cap, dict_size=ds)
    for k, s in specials.items():
        streams[k] = s
        w = oracle.lzma2_raw(s.data, s.dict_size, s.out_cap) if s.fmt == FMT_LZMA2_RAW else oracle.lzma1_alone(s.data, s.out_cap)
        want[k] = (w[0], w[1])
    assert sum(len(s.data) for s in streams) >= 64 << 20 and n >= 16384
    for _ in range(2):
        got = lzma_amd.decode_batch(ctx, streams)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g[1] == w[1], (i, g[1], w[1])
            assert g[0] == w[0], i


def test_descriptor_flags_and_reserved_bytes_are_checked(ctx):
    """`flags` was a reserved byte before round 4.  Unknown flag bits, the LZMA2 slice flag on a format that has
    no slices and non-zero reserved bytes fail the CALL with XLZ_ERR_BAD_ARG instead of silently changing what a stream
    means; a zeroed descriptor decodes."""
    import ctypes
    from lzma_amd import _native as N
    from lzma_amd import ERR_BAD_ARG
    p = corpus.plain("T", 77, 3000)
    c = corpus.compress_alone(p)
    buf = ctypes.create_string_buffer(c, len(c))
    out = ctypes.create_string_buffer(len(p))

    def call(flags=0, reserved=0, fmt=FMT_LZMA_ALONE):
        d = (N.StreamDesc * 1)()
        d[0].inp, d[0].in_len = ctypes.cast(buf, ctypes.c_void_p), len(c)
        d[0].out, d[0].out_cap = ctypes.cast(out, ctypes.c_void_p), len(p)
        d[0].format, d[0].flags = fmt, flags
        d[0].reserved[5] = reserved
        r = (N.Result * 1)()
        return N.lib().xlz_decode_batch(ctx._h, d, 1, r), r[0].status

    assert call() == (0, 0) and out.raw == p
    assert call(flags=1)[0] == ERR_BAD_ARG          # a slice of an LZMA1 stream does not exist
    assert call(flags=2)[0] == ERR_BAD_ARG          # no such flag
    assert call(reserved=7)[0] == ERR_BAD_ARG
    assert call(flags=1, fmt=lzma_amd.FMT_LZMA2_RAW)[0] == 0   # (the call runs; what the bytes decode to is another matter)
