"""Streams no compressor writes, built packet by packet with tests/lzma_craft.py: a match as the
very first packet (window.CheckDistance's off-by-one, window.go:89-91), rep matches and matched
literals that read window bytes of an EARLIER dictionary epoch after an LZMA2 dictionary reset
(window.Reset keeps the buffer, window.go:135-140), distances at the edge of the dictionary,
maximum-length matches, truncated first chunks behind stored chunks (reader2.go:146-153).

CPU: the crafter against liblzma (valid streams) and the oracle.  GPU: the HIP path against the
oracle on bytes, status and consumed input."""
import lzma
import random

import pytest

import oracle
from lzma_craft import (Encoder, Window, alone_header, lzma2_lzma_chunk, lzma2_stored, props_byte)


def _random_valid_stream(seed, lc, lp, pb, dict_size, n_packets):
    rnd = random.Random(seed)
    e = Encoder(lc, lp, pb, dict_size)
    for k in range(n_packets):
        fill = len(e.w.total)
        r = rnd.random()
        if fill == 0 or r < 0.35:
            e.literal(rnd.randrange(256) if rnd.random() < 0.5 else (fill * 7) & 0xFF)
        elif r < 0.65:
            d = rnd.choice([1, 2, 3, 5, 17, 200, 4000, fill, max(1, fill - 1), dict_size])
            d = max(1, min(d, fill, dict_size))
            e.match(d, rnd.choice([2, 3, 9, 10, 17, 18, 64, 65, 273]))
        elif r < 0.8:
            e.short_rep()
        else:
            e.rep(rnd.randrange(4), rnd.choice([2, 8, 9, 16, 20, 100, 273]))
    return e


def test_crafter_writes_valid_lzma_that_liblzma_and_the_oracle_decode():
    for seed, (lc, lp, pb, ds) in enumerate([(3, 0, 2, 65536), (0, 0, 0, 4096), (2, 1, 1, 8192), (4, 0, 4, 1 << 20),
                                             (0, 4, 0, 5000), (1, 1, 3, 4096)]):
        e = _random_valid_stream(seed, lc, lp, pb, ds, 700)
        want = bytes(e.w.total)
        e.end_marker()
        blob = alone_header(lc, lp, pb, ds) + e.payload()
        if ds % 16 == 0 or len(want) <= ds:  # else the reference's wrapped posState leaves the format (parity note 1)
            assert lzma.decompress(blob, format=lzma.FORMAT_ALONE) == want
        got = oracle.lzma1_alone(blob, len(want) + 10)
        if ds % 16 == 0 or len(want) <= ds:
            assert got == (want, 0, len(blob))


def crafted_lzma1():
    """(name, blob, out_cap) -- LZMA1 edge cases"""
    out = []
    # the very first packet is a match: distance 1 at position 0 passes CheckDistance (rep0 <= pos)
    for length in (2, 5, 64, 273):
        e = Encoder()
        e.match(1, length)
        e.literal(65)
        e.match(3, 7)
        e.end_marker()
        out.append(("first packet is a match of %d" % length, alone_header(3, 0, 2, 65536) + e.payload(), 1000))
    # ... the same with a defined size and no end marker
    e = Encoder()
    e.match(1, 9)
    e.literal(66)
    out.append(("first packet match, known size", alone_header(3, 0, 2, 65536, size=10) + e.payload(), 10))
    # first packet a match with distance 2: rejected (rep0 = 1 > pos = 0)
    e = Encoder()
    e.match(2, 4, copy=False)
    out.append(("first packet distance 2", alone_header(3, 0, 2, 65536) + e.payload() + bytes(8), 100))
    # first packet a rep match / short rep: window empty -> ErrResultError (decompress.go:690-692)
    e = Encoder()
    e.short_rep()
    out.append(("short rep on an empty window", alone_header(3, 0, 2, 65536) + e.payload(), 100))
    # distance == dictSize exactly once the window is full, and dictSize + 1 (rejected)
    for extra in (0, 1):
        e = Encoder(0, 0, 0, 4096)
        for i in range(4200):
            e.literal((i * 31) & 0xFF)
        e.match(4096 + extra, 20, copy=(extra == 0))
        e.literal(1)
        e.end_marker()
        out.append(("distance dictSize + %d" % extra, alone_header(0, 0, 0, 4096) + e.payload(), 5000))
    # maximum-length matches and reps back to back, overlapping (distance < length)
    e = Encoder(3, 0, 2, 1 << 16)
    for b in b"abcdefg":
        e.literal(b)
    for _ in range(40):
        e.match(7, 273)
        e.rep(0, 273)
        e.short_rep()
    e.end_marker()
    out.append(("long overlapping copies", alone_header(3, 0, 2, 1 << 16) + e.payload(), 40 * 547 + 100))
    return out


def crafted_lzma2():
    """(name, blob, dict_size, out_cap) -- LZMA2 streams whose copies cross a dictionary reset"""
    out = []
    for name, dict_size, first_len in (("unwritten", 1 << 16, 3000), ("short epoch", 4096, 4000),
                                       ("wrapped epoch", 4096, 10_000)):
        w = Window(dict_size)
        e = Encoder(3, 0, 2, dict_size, window=w)
        # chunk A (0xE0): epoch 1 -- text, then matches that leave four distinct reps behind
        rnd = random.Random(first_len)
        for i in range(first_len):
            e.literal(rnd.randrange(97, 123))
        e.match(min(first_len, dict_size) - 1, 5)
        e.match(77, 4)
        e.match(min(first_len, dict_size) // 2, 6)
        e.match(1500, 3)
        unc_a = len(w.total)
        blob = lzma2_lzma_chunk(0xE0, unc_a, e.payload(), props_byte(3, 0, 2))
        # stored chunk with a dictionary reset: epoch 2 starts, only 5 bytes in it
        w.reset()
        for b in b"RESET":
            w.put(b)
        blob += lzma2_stored(b"RESET", dict_reset=True)
        # chunk B (0x80: nothing reset): the model, the state (>= 7: the first literal is a MATCHED
        # literal whose matchByte lies behind the reset) and the four reps live on
        e.new_chunk()
        start = len(w.total)
        e.literal(0x41)
        e.rep(0, 8)       # rep0 = 1499: far behind the 5 + 1 bytes of this epoch -> stale window bytes
        e.literal(0x42)   # matched literal again
        e.rep(2, 30)
        e.short_rep()
        e.rep(3, 273)     # long: replicates stale bytes
        e.rep(1, 2)
        e.match(3, 4)     # a NEW distance inside the epoch is fine
        e.rep(1, 100)
        blob += lzma2_lzma_chunk(0x80, len(w.total) - start, e.payload())
        # another reset and a third epoch that reads bytes of epoch 2 AND (where epoch 2 was short) epoch 1
        w.reset()
        for b in b"xy":
            w.put(b)
        blob += lzma2_stored(b"xy", dict_reset=True)
        e.new_chunk()
        start = len(w.total)
        e.rep(0, 50)
        e.rep(2, 273)
        e.literal(0x43)
        blob += lzma2_lzma_chunk(0x80, len(w.total) - start, e.payload())
        blob += b"\x00"
        out.append((name, blob, dict_size, len(w.total) + 100, bytes(w.total)))
    return out


def crafted_lzma2_cut_chunks():
    """(name, blob, dict_size, out_cap) -- a compressed chunk whose input ends INSIDE a packet, followed by
    a chunk that resets nothing (0x80).  The reference mutates state and reps before the ReadByte that
    fails (decompress.go:216 shifts the reps before the length is read, :431 moves the state before the
    distance, :785-798 rotate the reps before the normalisation), returns io.EOF, and Reader2 starts the
    next chunk on whatever was left: its first literal is a matched literal at the rep0 of that moment.
    (Found by tools/fuzz_gpu.py seed 424242; tests/golden/fuzz_424242_45182.lzma2 is that input.)"""
    out = []
    ds = 1 << 16
    for tail in range(6):
        rnd = random.Random(500 + tail)
        w = Window(ds)
        e = Encoder(3, 0, 2, ds, window=w)
        for i in range(2500):
            e.literal(rnd.randrange(97, 105))
        for _ in range(40):
            e.match(rnd.randrange(1, 2400), rnd.choice([2, 5, 30]))
            e.literal(rnd.randrange(97, 105))
        # the packets the cut lands in: one kind per variant, repeated so that every cut of 1..14 bytes
        # ends inside one of them
        for _ in range(8):
            if tail == 0:
                e.match(rnd.randrange(1200, 2400), 9)      # distance with direct bits + align
            elif tail == 1:
                e.rep(1, 4)                                # reps rotated before the normalisation
            elif tail == 2:
                e.rep(2, 17)
            elif tail == 3:
                e.rep(3, 70)
            elif tail == 4:
                e.match(rnd.randrange(1, 100), 273)        # long length, short distance
                e.short_rep()
            else:
                e.match(rnd.randrange(300, 2000), 3)
                e.literal(rnd.randrange(256))              # matched literal
        pay_a, n_a = e.payload(), len(w.total)
        e.new_chunk()
        for _ in range(60):
            e.literal(rnd.randrange(97, 105))
            e.rep(rnd.randrange(4), rnd.choice([2, 9]))
            e.match(rnd.randrange(1, 2400), rnd.choice([2, 3, 8]))
        pay_b, n_b = e.payload(), len(w.total) - n_a
        chunk_b = lzma2_lzma_chunk(0x80, n_b, pay_b)
        for cut in range(1, 15):
            blob = lzma2_lzma_chunk(0xE0, n_a, pay_a[:-cut], props_byte(3, 0, 2)) + chunk_b + b"\x00"
            out.append(("variant %d, chunk cut %d bytes short" % (tail, cut), blob, ds, n_a + n_b + 64))
    return out


def test_cut_chunks_carry_half_done_packets_cpu():
    """the oracle on the cut chunks: the second chunk starts where the first stopped (less output than the
    header promised is NOT an error in the reference), and what follows depends on the cut"""
    seen = set()
    for name, blob, ds, cap in crafted_lzma2_cut_chunks():
        out, status, consumed = oracle.lzma2_raw(blob, ds, cap)
        seen.add((status, len(out)))
        assert consumed > 6, name
    assert len(seen) > 20
    blob, ds, cap = _fuzz_find()
    out, status, consumed = oracle.lzma2_raw(blob, ds, cap)
    assert (status, len(out), consumed) == (oracle.ERR_RESULT, 9942, 907)


def _fuzz_find():
    """the differential fuzzer's find: a bit flip made the chunk at offset 817 run out of input inside a
    match (after the reps were shifted, before the distance was read)"""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_424242_45182.lzma2")
    return open(path, "rb").read(), 8192, 10671


def _fuzz_find_wide():
    """round 4's find (tools/fuzz_gpu.py seed 5501): a damaged stream whose real decode leaves its chunk headers and walks
    into bytes that read as a chunk with LARGER properties than any header the host's scan saw.  The model's storage is sized
    by that scan, so the wave stopped with "unsupported" where the reference renews its model (reader2.go:155-165) and
    decodes on: 9481 bytes, no error.  Such a stream is now decoded once more with room for the largest model."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_5501_18774.lzma2")
    return open(path, "rb").read(), 8192, 9481


def walks_into_larger_pb():
    """the same with properties that need no larger MODEL but a larger LAYOUT (round 5): the hidden chunk renews the model
    with lc 0 / lp 0 / pb 4 -- sixteen posStates, where the launch, whose units' headers all said pb <= 2, laid its model
    out with room for four (xlz_format.h: ModelLayout<true>).  The unit stops in front of that chunk (AUX_GROW) and the
    stream is decoded again by the widest launch, which uses the full layout; the reference simply decodes on
    (reader2.go:155-165).  Literals at posStates 4..15 would index beyond the compact tables."""
    e = Encoder()
    for b in b"abcabcabcabc":
        e.literal(b)
    p1 = e.payload()
    e2 = Encoder(lc=0, lp=0, pb=4)
    for b in b"the quick brown fox jumps over the lazy dog":
        e2.literal(b)
    e2.match(9, 20)
    second = lzma2_lzma_chunk(0xE0, 43 + 20, e2.payload(), props_byte(0, 0, 4))
    body = p1 + second
    hdr = bytes([0xE0, 0, 11]) + (len(body) + 7 - 1).to_bytes(2, "big") + bytes([props_byte(3, 0, 2)])
    return hdr + body + b"\x00" * 8, 1 << 16, 200


def walks_into_larger_props():
    """the same on purpose.  A chunk ends when its announced output is there (decompress.go:14-20); what its header
    announced as compressed size beyond that stays in the source, and startChunk reads the next control byte from
    THERE (reader2.go:100-128: the limitedByteReader only caps the chunk's reads).  The first chunk here announces more
    compressed bytes than its twelve literals need, so the host's header scan jumps over what follows -- a chunk that
    resets the dictionary and renews the model with lc 8 / lp 4 (a model of 6 MiB) -- while the real decode walks
    into it."""
    e = Encoder()
    for b in b"abcabcabcabc":
        e.literal(b)
    p1 = e.payload()
    e2 = Encoder(lc=8, lp=4, pb=0)
    for b in b"xyzzy":
        e2.literal(b)
    second = lzma2_lzma_chunk(0xE0, 5, e2.payload(), props_byte(8, 4, 0))
    body = p1 + second
    hdr = bytes([0xE0, 0, 11]) + (len(body) + 7 - 1).to_bytes(2, "big") + bytes([props_byte(3, 0, 2)])
    return hdr + body + b"\x00" * 8, 1 << 16, 100


def crafted_lzma2_framing():
    """first LZMA chunk behind stored chunks: Reader2.lzmaReader is still nil there, so an EOF inside
    rangeDec.Init is a constructor error (reader2.go:146-153)"""
    e = Encoder()
    for b in b"hello hello hello":
        e.literal(b)
    pay = e.payload()
    good = lzma2_lzma_chunk(0xE0, 17, pay, props_byte(3, 0, 2))
    out = []
    for cut in (1, 2, 4, 5, 6, 8, 9, len(good) - 1):
        out.append(("stored, then LZMA chunk cut at %d" % cut, lzma2_stored(b"abc", True) + good[:cut], 1 << 16, 100))
        out.append(("stored, LZMA, then LZMA chunk cut at %d" % cut, lzma2_stored(b"abc", True) + good + good[:cut], 1 << 16, 100))
    return out


def test_crafted_streams_cpu_expectations():
    """the crafter's own window model agrees with the oracle on what the crafted LZMA2 streams
    decode to (two independent restatements of window.go's uncleared Reset)"""
    for name, blob, ds, cap, want in crafted_lzma2():
        got = oracle.lzma2_raw(blob, ds, cap)
        assert got[1] == 0, name
        assert got[0] == want, name
        # the stale bytes matter: after the reset the copies brought back lower-case text of epoch 1
        # (a decoder that clears the window on Reset, or reads zeros there, gives other bytes)
        tail = want[want.index(b"RESET") + 5:]
        if name != "unwritten":  # (there the reads land where no epoch ever wrote: zeros)
            assert sum(1 for b in tail if 97 <= b <= 122) > 100, name
    for name, blob, cap in crafted_lzma1():
        oracle.lzma1_alone(blob, cap)  # must not crash; statuses are compared on the GPU
    # a real decode that walks into a chunk the header scan jumps over (larger properties: round 4's fuzzer find)
    blob, ds, cap = walks_into_larger_props()
    assert oracle.lzma2_raw(blob, ds, cap) == (b"abcabcabcabcxyzzy", 0, 39)
    blob, ds, cap = walks_into_larger_pb()
    fox = b"the quick brown fox jumps over the lazy dog"
    got = oracle.lzma2_raw(blob, ds, cap)
    assert got[:2] == (b"abcabcabcabc" + fox + (fox[-9:] * 3)[:20], 0)
    import lzma_pydec
    assert lzma_pydec.lzma2_raw(blob, ds)[:2] == got[:2]
    import lzma_amd
    assert [u["out_len"] for u in lzma_amd.lzma2_units(blob)] == [12]
    blob, ds, cap = _fuzz_find_wide()
    out, status, consumed = oracle.lzma2_raw(blob, ds, cap)
    assert (status, len(out), consumed) == (0, 9481, 930)


def test_random_crafted_lzma2_streams_cpu():
    """random packets, random chunk kinds and resets (tools/fuzz_gpu.py uses the same generator on the
    GPU): the crafter's window model and the oracle agree on every byte"""
    from lzma_craft import random_lzma2_stream
    for seed in range(150):
        rnd = random.Random(seed)
        ds = rnd.choice([4096, 4097, 8192, 65536])
        blob, want = random_lzma2_stream(rnd, ds)
        got = oracle.lzma2_raw(blob, ds, len(want) + 100)
        assert got[1] == 0 and got[0] == want, seed


def test_long_first_epoch_and_large_model_streams_cpu():
    """the generators behind the session tests (tests/test_gpu_readers.py): a stream whose first dictionary epoch is
    megabytes long with reads behind later resets, and random LZMA2 streams whose chunks renew the model with lc+lp up
    to 12 -- crafter's window model = oracle (= the third restatement on a short one)"""
    import lzma_pydec
    from lzma_craft import ANY_PROPS, long_stale_lzma2_stream, random_lzma2_stream
    for ds in (4096, 65536):
        blob, want = long_stale_lzma2_stream(ds, first_epoch=300_000, seed=ds)
        assert oracle.lzma2_raw(blob, ds, len(want) + 100) == (want, 0, len(blob))
    blob, want = long_stale_lzma2_stream(4096, first_epoch=30_000)
    assert lzma_pydec.lzma2_raw(blob, 4096)[:2] == (want, 0)
    seen = set()
    for seed in range(7000, 7060):
        rnd = random.Random(seed)
        ds = rnd.choice([4096, 4097, 65536])
        blob, want = random_lzma2_stream(rnd, ds, max_chunks=10, props=ANY_PROPS)
        assert oracle.lzma2_raw(blob, ds, len(want) + 100) == (want, 0, len(blob)), seed
        pos = 0
        while blob[pos] != 0:           # the properties the chunks bring
            c = blob[pos]
            if c >= 0x80:
                if c >= 0xC0:
                    seen.add((blob[pos + 5] % 9) + (blob[pos + 5] // 9) % 5)
                pos += (6 if c >= 0xC0 else 5) + ((blob[pos + 3] << 8) | blob[pos + 4]) + 1
            else:
                pos += 3 + ((blob[pos + 1] << 8) | blob[pos + 2]) + 1
    assert max(seen) == 12 and {5, 6, 7, 8}.intersection(seen) and min(seen) <= 4


@pytest.mark.gpu
def test_random_crafted_lzma2_streams_gpu(ctx):
    import lzma_amd
    from lzma_amd import FMT_LZMA2_RAW, Stream
    from lzma_craft import random_lzma2_stream
    streams, wants = [], []
    for seed in range(1000, 1400):
        rnd = random.Random(seed)
        ds = rnd.choice([4096, 4097, 8192, 65536])
        blob, want = random_lzma2_stream(rnd, ds)
        cap = len(want) + rnd.choice([0, 0, 9, -1])
        streams.append(Stream(blob, FMT_LZMA2_RAW, out_cap=max(cap, 0), dict_size=ds))
        wants.append(oracle.lzma2_raw(blob, ds, max(cap, 0)))
    got = lzma_amd.decode_batch(ctx, streams)
    for i, (g, w) in enumerate(zip(got, wants)):
        assert g == w, i


@pytest.mark.gpu
def test_crafted_streams_on_gpu(ctx):
    import lzma_amd
    from lzma_amd import FMT_LZMA2_RAW, FMT_LZMA_ALONE, Stream
    c1 = crafted_lzma1()
    got = lzma_amd.decode_batch(ctx, [Stream(b, FMT_LZMA_ALONE, out_cap=cap) for _, b, cap in c1])
    for (name, b, cap), g in zip(c1, got):
        assert g == oracle.lzma1_alone(b, cap), name
    c2 = [(n, b, ds, cap) for n, b, ds, cap, _ in crafted_lzma2()] + crafted_lzma2_framing() + crafted_lzma2_cut_chunks()
    # alone, and inside a batch next to ordinary streams (the exact re-run must not disturb them)
    import corpus
    p = corpus.plain("T", 77, 50_000)
    extra = Stream(corpus.compress_raw_lzma2(p), FMT_LZMA2_RAW, out_cap=len(p), dict_size=1 << 16)
    c2.append(("fuzz find 424242/45182",) + _fuzz_find())
    c2.append(("fuzz find 5501/18774",) + _fuzz_find_wide())
    c2.append(("off the headers into lc 8 / lp 4",) + walks_into_larger_props())
    c2.append(("off the headers into pb 4 (a compact-layout launch)",) + walks_into_larger_pb())
    streams = [Stream(b, FMT_LZMA2_RAW, out_cap=cap, dict_size=ds) for _, b, ds, cap in c2]
    got = lzma_amd.decode_batch(ctx, streams + [extra])
    for (name, b, ds, cap), g in zip(c2, got):
        assert g == oracle.lzma2_raw(b, ds, cap), name
    assert got[-1][0] == p
    # the same streams through the pull readers (session -> whole-stream fallback for the stale ones)
    for name, b, ds, cap in c2:
        want = oracle.lzma2_raw(b, ds, cap)
        r, err = lzma_amd.NewReader2(ctx, b, ds)
        if r is None:
            continue  # constructor errors are covered by test_reader_constructor_errors
        out, e = r.read_all(chunk=1000)
        assert out == want[0], name
        assert (e is None) == (want[1] >= 0), name
