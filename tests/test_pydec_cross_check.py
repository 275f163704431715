"""CPU: the C oracle against a third, independent restatement in plain Python (tests/lzma_pydec.py) on the
LZMA2 cases where the reference's behaviour is least obvious: chunks cut inside a packet (what the next
chunk inherits), stale window bytes behind dictionary resets, damaged framing, the fuzzers' finds."""
import os
import random

import oracle
import lzma_pydec
from lzma_craft import random_lzma2_stream
from test_crafted_streams import crafted_lzma2, crafted_lzma2_cut_chunks, crafted_lzma2_framing

HERE = os.path.dirname(os.path.abspath(__file__))


def _same(blob, ds, name):
    want = oracle.lzma2_raw(blob, ds, 1 << 22)
    got = lzma_pydec.lzma2_raw(blob, ds)
    assert got == want, (name, got[1:], want[1:], len(got[0]), len(want[0]))


def test_cut_chunks_and_stale_windows():
    for name, blob, ds, cap in crafted_lzma2_cut_chunks():
        _same(blob, ds, name)
    for name, blob, ds, cap, _ in crafted_lzma2():
        _same(blob, ds, name)
    for name, blob, ds, cap in crafted_lzma2_framing():
        _same(blob, ds, name)


def test_fuzzer_finds():
    for fn, ds in (("fuzz_424242_45182.lzma2", 8192), ("fuzz_reader_11_316.lzma2", 65536), ("fuzz_5501_18774.lzma2", 8192)):
        _same(open(os.path.join(HERE, "golden", fn), "rb").read(), ds, fn)


def test_random_crafted_streams_with_bit_flips():
    n_err = 0
    for seed in range(3000, 3120):
        rnd = random.Random(seed)
        ds = rnd.choice([4096, 4097, 8192, 65536])
        blob, _ = random_lzma2_stream(rnd, ds, max_chunks=6, max_packets=60)
        if seed % 2:
            b = bytearray(blob)
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
            blob = bytes(b)
        _same(blob, ds, seed)
        n_err += oracle.lzma2_raw(blob, ds, 1 << 22)[1] != 0
    assert n_err > 10


def test_lzma1_assets_and_every_truncation_of_one():
    """the reference's small .lzma assets (good and bad ones), and a.lzma cut at every length: constructor
    errors, the clean io.EOF of a truncated stream, end marker with and without a known size"""
    g = os.path.join(HERE, "golden")
    for fn in sorted(os.listdir(g)):
        if fn.endswith(".lzma") and os.path.getsize(os.path.join(g, fn)) < 4096:
            blob = open(os.path.join(g, fn), "rb").read()
            assert lzma_pydec.lzma1_alone(blob) == oracle.lzma1_alone(blob, 1 << 20), fn
    blob = open(os.path.join(g, "a.lzma"), "rb").read()
    for cut in range(len(blob)):
        assert lzma_pydec.lzma1_alone(blob[:cut]) == oracle.lzma1_alone(blob[:cut], 1 << 20), cut
    for fn in ("a_eos.lzma", "a_eos_and_size.lzma", "bad_eos_incorrect_size.lzma"):
        blob = open(os.path.join(g, fn), "rb").read()
        for cut in range(13, len(blob), 7):
            assert lzma_pydec.lzma1_alone(blob[:cut]) == oracle.lzma1_alone(blob[:cut], 1 << 20), (fn, cut)


def test_damaged_liblzma_streams_lzma1_and_lzma2():
    """real compressor output (liblzma), small, with truncations and bit flips: the error paths of both
    restatements -- distances beyond the window, rep matches on an empty window, markers with a non-zero
    code, sizes that do not add up -- agree"""
    import corpus
    rnd = random.Random(77)
    n_bad = 0
    for k in range(140):
        fam = "TRMZ"[k % 4]
        p = corpus.plain(fam, 9000 + k, rnd.choice([1, 40, 700, 2500]))
        lc = rnd.randrange(0, 5)
        lp = rnd.randrange(0, 5 - lc)
        pb = rnd.randrange(0, 5)
        if k % 3:
            blob = bytearray(corpus.compress_alone(p, dict_size=rnd.choice([4096, 5000, 65536]), lc=lc, lp=lp, pb=pb,
                                                   known_size=bool(k % 2), preset=0))
            if k % 5:
                for _ in range(rnd.randrange(1, 3)):
                    blob[rnd.randrange(13 if k % 7 else 0, len(blob))] ^= 1 << rnd.randrange(8)
            if k % 11 == 0:
                blob = blob[: rnd.randrange(1, len(blob))]
            blob = bytes(blob)
            want = oracle.lzma1_alone(blob, 1 << 22)
            if want[1] == oracle.ERR_OUT_CAP:
                continue
            assert lzma_pydec.lzma1_alone(blob) == want, k
        else:
            ds = rnd.choice([4096, 65536])
            blob = bytearray(corpus.lzma2_concat([p, p[: len(p) // 2] or p], dict_size=ds, lc=lc, lp=lp, pb=pb))
            if k % 2:
                blob[rnd.randrange(len(blob))] ^= 1 << rnd.randrange(8)
            blob = bytes(blob)
            want = oracle.lzma2_raw(blob, ds, 1 << 22)
            if want[1] == oracle.ERR_OUT_CAP:
                continue
            assert lzma_pydec.lzma2_raw(blob, ds) == want, k
        n_bad += want[1] < 0
    assert n_bad > 15
