"""GPU: the BASELINE shapes that no oracle-sized test reaches: a 65 536-entry
work queue, an output arena beyond 4 GiB (64-bit arena offsets), and one stream whose 8 MiB window
wraps with distances up to the dictionary size (window.go:31-87, decompress.go:22,56,651-653)."""
import hashlib

import pytest

import corpus
import lzma_amd
import oracle
from lzma_amd import FMT_LZMA_ALONE, Stream

pytestmark = pytest.mark.gpu


def test_65536_streams_through_one_batch(ctx):
    """cfg3's stream count (8 KiB each so that the corpus builds in seconds): SHA-256 of every
    stream, per-stream results, and the batch's byte accounting."""
    n, size = 65536, 8192
    comp, digests = corpus.make_alone_batch("M", n, size, base_seed=90_000, preset={"mode": 1, "mf": 3, "nice_len": 32,
                                                                                    "depth": 2})
    b = lzma_amd.Batch(ctx, [Stream(c, FMT_LZMA_ALONE, out_cap=size) for c in comp])
    b.run()
    res = b.results()
    assert len(res) == n and all(r[1] == 0 and r[0] == size for r in res)
    assert all(r[2] == len(c) for r, c in zip(res, comp))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(lambda i: hashlib.sha256(b.download(i, size)).digest(), range(n)))
    assert got == digests
    cin, cout, units = b.stats()
    assert cout == n * size and units == n and cin == sum(len(c) - 13 for c in comp)
    b.close()


def test_output_arena_beyond_4_gib(ctx):
    """4352 streams x 1 MiB: stream regions start past offset 2^32 in the output arena (and the
    input arena offsets are 64-bit too); 64 distinct plaintexts, every region checked."""
    nd, n, size = 64, 4352, 1 << 20
    ps = [corpus.plain("TMZR"[i % 4], 91_000 + i, size) for i in range(nd)]
    cs = [corpus.compress_alone(p, preset=0) for p in ps]
    hs = [hashlib.sha256(p).digest() for p in ps]
    b = lzma_amd.Batch(ctx, [Stream(cs[(i * 7) % nd], FMT_LZMA_ALONE, out_cap=size) for i in range(n)])
    ptr0, _ = b.device_output(0)
    ptr_last, _ = b.device_output(n - 1)
    assert ptr_last - ptr0 > (1 << 32)
    b.run()
    res = b.results()
    assert all(r[1] == 0 and r[0] == size for r in res)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        ok = list(ex.map(lambda i: hashlib.sha256(b.download(i, size)).digest() == hs[(i * 7) % nd], range(n)))
    assert all(ok)
    b.close()


def test_one_stream_wraps_an_8_mib_window(ctx):
    """20 MiB, props 0x38 (lc2 lp1 pb1, BASELINE config 5), 8 MiB dictionary, repeats 6-8 MiB back:
    window.pos wraps twice and the copies reach almost a whole dictionary back; byte for byte and
    on status / consumed input against the oracle.  A second stream with a dictionary size that is
    not a multiple of 16 (the wrapped position feeds posState and the literal context)."""
    p = corpus.plain("F", 92_000, 20 << 20)
    c = corpus.compress_alone(p, dict_size=8 << 20, lc=2, lp=1, pb=1, preset=0)
    assert c[0] == 0x38
    p2 = corpus.plain_far(92_001, 3 << 20, far_lo=900_000, far_hi=1_000_000)
    c2 = bytearray(corpus.compress_alone(p2, dict_size=1 << 20, lc=2, lp=1, pb=1, preset=0))
    c2[1:5] = (1_048_583).to_bytes(4, "little")  # window.size not a multiple of 16 (and >= the encoder's 1 MiB)
    got = lzma_amd.decode_batch(ctx, [Stream(c, FMT_LZMA_ALONE, out_cap=len(p)), Stream(bytes(c2), FMT_LZMA_ALONE,
                                                                                         out_cap=len(p2))])
    want = [oracle.lzma1_alone(c, len(p)), oracle.lzma1_alone(bytes(c2), len(p2))]
    assert got[0][1] == want[0][1] == 0 and got[0][2] == want[0][2]
    assert got[0][0] == want[0][0] == p
    # window.size % 16 != 0: after the first wrap the reference's posState (wrapped window.pos,
    # decompress.go:22) leaves the encoder's, the decode derails -- identically in the oracle
    assert len(want[1][0]) > 1_048_583 and want[1][0][:1_048_583] == p2[:1_048_583]
    assert got[1] == want[1]


@pytest.mark.parametrize("n,size,pieces,sliced", [(12288, 256 << 10, 3, True), (16384, 128 << 10, 5, False)])
def test_a_call_of_several_wave_rounds_runs_as_a_pipeline(ctx, n, size, pieces, sliced):
    """xlz_decode_batch with more streams than the chip holds at once, host buffers in and out, in its two forms
    (xlz_call_stats.sub_batches / slices).  12 288 streams of 256 KiB: LONG streams -- a wave round of them is a GiB -- run
    as three pieces of exactly one round each, one behind the other, each overlapping its copies with its own decode
    (slices).  16 384 streams of 128 KiB: five pieces (a quarter-size one in front and behind, three whole ones) whose
    upload, decode and download overlap and whose launches follow each other on two streams.  Every stream's bytes, status
    and consumed input as for the same streams decoded alone -- including damaged streams in every piece (one bad stream
    never fails the call) and an empty one."""
    import ctypes
    import numpy as np
    from lzma_amd import _native as N
    nd = 48
    ps = [corpus.plain("TMZR"[i % 4], 93_000 + i, size) for i in range(nd)]
    cs = [corpus.compress_alone(p, preset=0) for p in ps]
    bad = {}
    for j in (5, 4100, 4101, 9000, 11000, n - 1):      # damaged: a flipped byte deep inside, or cut short
        c = bytearray(cs[j % nd])
        if j % 2:
            c[len(c) // 2] ^= 0x55
        else:
            del c[len(c) * 2 // 3:]
        bad[j] = bytes(c)
    bad[7000] = b""
    ins = [np.frombuffer(bad.get(i, cs[i % nd]), dtype=np.uint8) for i in range(n)]
    out = np.zeros((n, size), dtype=np.uint8)
    descs = (N.StreamDesc * n)()
    for i in range(n):
        descs[i].inp = ins[i].ctypes.data if ins[i].size else None
        descs[i].in_len = ins[i].size
        descs[i].out, descs[i].out_cap = out[i].ctypes.data, size
        descs[i].format = FMT_LZMA_ALONE
    res = (N.Result * n)()
    assert N.lib().xlz_decode_batch(ctx._h, descs, n, res) == 0
    st = ctx.last_call_stats()
    assert st["sub_batches"] == pieces and st["streams"] == n and st["units"] == n - 1   # (the empty stream has no unit)
    assert (st["slices"] >= 4) == sliced and 0.3 < st["slot_occupancy"] <= 1.0 and st["total_ms"] > 0
    hs = [hashlib.sha256(p).digest() for p in ps]
    for i in range(n):
        if i in bad:
            want = oracle.lzma1_alone(bad[i], size)
            assert (bytes(out[i][: res[i].out_len]), res[i].status, res[i].in_consumed) == want, i
        else:
            assert res[i].status == 0 and res[i].out_len == size and res[i].in_consumed == len(cs[i % nd]), i
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        ok = list(ex.map(lambda i: i in bad or hashlib.sha256(out[i]).digest() == hs[i % nd], range(n)))
    assert all(ok)
    # the same call again (pinned pools warm), and a small call right after it: one sub-batch
    assert N.lib().xlz_decode_batch(ctx._h, descs, n, res) == 0
    assert all(res[i].status == 0 for i in range(n) if i not in bad)
    got = lzma_amd.decode_batch(ctx, [Stream(cs[0], FMT_LZMA_ALONE, out_cap=size)])
    assert got[0][0] == ps[0] and ctx.last_call_stats()["sub_batches"] == 1


def test_a_call_of_one_wave_round_runs_in_slices(ctx):
    """xlz_decode_batch's sliced form (xlz_call_stats.slices; forced on a small call with xlz_ctx_set_slicing): the call is
    a sequence of launches that each advance every unit by a share of its output, and share k - 1 goes to the callers'
    buffers while share k decodes -- the reference's Read pump in batch form (reader1.go:223-254, window.go:97-133).
    Streams of every kind in ONE call, against the oracle on bytes, status and consumed input: all four plaintext
    families at several sizes, known sizes without end marker, output room that is too small, streams cut short or
    with a flipped byte, an empty one, tiny ones (shorter than a slice's 256-byte grain), LZMA2 streams of several
    units, of stored chunks and damaged, crafted LZMA2 streams whose copies read behind dictionary resets (settled
    by the exact re-run AFTER their slices have gone out: those bytes are fetched again), and streams whose head
    compresses so badly that the first launch -- which starts on a share of every input -- falls short of its bound."""
    import random
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import lzma_craft
    from lzma_amd import FMT_LZMA2_RAW
    rnd = random.Random(5005)
    jobs = []  # (Stream, oracle call)

    def alone(c, cap):
        jobs.append((Stream(c, FMT_LZMA_ALONE, out_cap=cap), lambda c=c, cap=cap: oracle.lzma1_alone(c, cap)))

    def raw2(c, cap, dict_size=65536):
        jobs.append((Stream(c, FMT_LZMA2_RAW, out_cap=cap, dict_size=dict_size),
                     lambda c=c, cap=cap, d=dict_size: oracle.lzma2_raw(c, d, cap)))

    for i in range(160):
        size = rnd.choice([300, 5_000, 70_000, 200_000, 333_333])
        p = corpus.plain("TMZR"[i % 4], 95_000 + i, size)
        c = corpus.compress_alone(p, preset=0, known_size=(i % 5 == 0))
        kind = i % 8
        if kind == 5:
            c = bytearray(c)
            c[13 + (len(c) - 13) * rnd.randrange(1, 9) // 10] ^= 1 << rnd.randrange(8)
            c = bytes(c)
        elif kind == 6:
            c = c[: 13 + (len(c) - 13) * rnd.randrange(1, 9) // 10]
        alone(c, size if kind != 7 else size * rnd.randrange(1, 9) // 10)   # kind 7: not enough room
    for i in range(6):   # an incompressible head, then long repeats: the first launch sees a share of the INPUT (the rest
        # is still being uploaded), runs out of it far in front of its output bound and pauses -- the bytes up to the bound
        # come out of a later launch, when that bound's pieces have gone out: the stream is fetched again
        p = corpus.plain("R", 95_400 + i, 60_000 + 10_000 * i) + corpus.plain("Z", 95_410 + i, 400_000)
        alone(corpus.compress_alone(p, preset=0), len(p))
    for i in range(12):                                                         # the a.lzma flavour
        p = corpus.plain("T", 95_500 + i, 60_000)
        c = corpus.alone_known_size_no_eos(p)
        if c:
            alone(c, len(p))
    alone(b"", 100)
    alone(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "a.lzma"), "rb").read(), 4096)
    for i in range(24):                                                         # LZMA2: units, stored chunks, damage
        segs = [corpus.plain("TRMZ"[(i + j) % 4], 96_000 + 10 * i + j, rnd.choice([40_000, 150_000, 262_144]))
                for j in range(rnd.randrange(1, 6))]
        c = corpus.lzma2_concat(segs, preset=0)
        total = sum(len(s) for s in segs)
        if i % 4 == 1:
            c = bytearray(c)
            c[len(c) * rnd.randrange(1, 9) // 10] ^= 0x10
            c = bytes(c)
        elif i % 4 == 2:
            c = c[: len(c) * rnd.randrange(3, 9) // 10]
        raw2(c, total if i % 6 else total - 1000)
    for i in range(40):                                                         # copies behind dictionary resets
        c, want = lzma_craft.random_lzma2_stream(rnd, dict_size=4096)
        raw2(c, len(want) + 64, dict_size=4096)
    beyond_lds = lzma_craft.SMALL_PROPS + [(8, 4, 2), (6, 4, 0), (8, 1, 0)]     # models beyond LDS (lc + lp = 9 .. 12): their own
    for i in range(16):                                                         # launch behind the slices, fetched at the end
        c, want = lzma_craft.random_lzma2_stream(rnd, dict_size=4096, props=beyond_lds)
        raw2(c, len(want) + 64, dict_size=4096)
    ctx.set_slicing(1, 1 << 20, 5)
    try:
        got = lzma_amd.decode_batch(ctx, [j[0] for j in jobs])
        st = ctx.last_call_stats()
        assert st["slices"] == 6 and st["sub_batches"] == 1 and 0 < st["slot_occupancy"] <= 1.0   # (five shares, the last one cut in two)
        for i, (_, want) in enumerate(jobs):
            assert got[i] == want(), i
        # two slices, and the plain call: the same results
        ctx.set_slicing(1, 1 << 20, 2)
        assert lzma_amd.decode_batch(ctx, [j[0] for j in jobs]) == got and ctx.last_call_stats()["slices"] == 2
        ctx.set_slicing(0, 0, 1)
        assert lzma_amd.decode_batch(ctx, [j[0] for j in jobs]) == got and ctx.last_call_stats()["slices"] <= 1
        # the context keeps the calls' device memory for the next call of the same shape (xlz_ctx_trim gives it back)
        assert ctx.trim() > 0 and ctx.trim() == 0
        assert lzma_amd.decode_batch(ctx, [j[0] for j in jobs]) == got
    finally:
        ctx.set_slicing(0, 0, 0)


def test_a_call_of_5000_streams_is_one_round_of_20_per_cu(ctx):
    """round 5: a launch that fits ONE round at a higher occupancy takes it -- 5000 units are one round of 20 workgroups
    per CU, not a full round of 16 per CU and a second one a fifth full -- and a call of one round runs in slices
    (xlz_call_stats: wave_slots, slices); every stream's bytes by SHA-256, a damaged one against the oracle."""
    n, size = 5000, 8192
    comp, digests = corpus.make_alone_batch("M", n, size, base_seed=97_000, preset={"mode": 1, "mf": 3, "nice_len": 32, "depth": 2})
    comp = list(comp)
    comp[1234] = comp[1234][: len(comp[1234]) // 2]
    ctx.set_slicing(1, 8 << 20, 4)
    try:
        got = lzma_amd.decode_batch(ctx, [Stream(c, FMT_LZMA_ALONE, out_cap=size) for c in comp])
        st = ctx.last_call_stats()
    finally:
        ctx.set_slicing(0, 0, 0)
    assert st["slices"] == 5 and st["sub_batches"] == 1 and st["units"] == n   # (four shares, the last one cut in two)
    assert st["wave_slots"] == n and 4096 < n <= 20 * 256      # (the grid is the units: all of them resident at once)
    assert got[1234] == oracle.lzma1_alone(comp[1234], size)
    assert all(g[1] == 0 and hashlib.sha256(g[0]).digest() == d for i, (g, d) in enumerate(zip(got, digests)) if i != 1234)


def test_concurrent_calls_on_one_context(ctx):
    """xlz_decode_batch is thread-safe across calls on ONE context (include/xlz.h): the pinned input image is leased to one
    call at a time (a sliced call keeps it until the tails of its inputs have left it), the pinned output ring likewise,
    the context's memory pool and its two launch streams are shared.  Four threads -- two sliced calls of one wave round,
    two small plain calls, three times each -- and every result as for the same streams decoded alone."""
    import threading
    sets = []
    for t in range(4):
        n, size = (600, 64 << 10) if t < 2 else (40, 20_000)
        ps = [corpus.plain("TMZR"[(i + t) % 4], 98_000 + 1000 * t + i, size) for i in range(24)]
        cs = [corpus.compress_alone(p, preset=0) for p in ps]
        streams = [Stream(cs[i % 24] if (i + t) % 17 else cs[i % 24][: len(cs[i % 24]) // 2], FMT_LZMA_ALONE, out_cap=size) for i in range(n)]
        sets.append(streams)
    want = []
    ctx.set_slicing(0, 0, 1)
    for s in sets:
        want.append(lzma_amd.decode_batch(ctx, s))      # alone, unsliced
    assert all(w[1] in (0, 1) and len(w[0]) > 0 for ws in want for w in ws)
    ctx.set_slicing(1, 4 << 20, 6)                       # (calls of 4 MiB and more run in slices: the first two sets)
    errors = []

    def work(k):
        try:
            for _ in range(3):
                got = lzma_amd.decode_batch(ctx, sets[k])
                assert got == want[k], "thread %d" % k
        except BaseException as e:   # noqa: BLE001 -- reported below
            errors.append(e)
    try:
        th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    finally:
        ctx.set_slicing(0, 0, 0)
    assert not errors, errors[0]


def test_the_context_keeps_what_the_last_calls_used_and_no_more(ctx):
    """xlz_decode_batch keeps its device and pinned memory in the context between calls (MemPool): a call of the same shape
    finds its blocks again, and what two calls in a row did not touch goes back to the system -- a context does not grow with
    every shape it has ever seen (xlz_ctx_trim says how much is kept and releases it)."""
    big = [Stream(corpus.compress_alone(corpus.plain("T", 99_000 + i, 1 << 20), preset=0), FMT_LZMA_ALONE, out_cap=1 << 20) for i in range(96)]
    small = [Stream(corpus.compress_alone(corpus.plain("T", 99_500 + i, 20_000), preset=0), FMT_LZMA_ALONE, out_cap=20_000) for i in range(8)]
    ctx.trim()
    want_big, want_small = lzma_amd.decode_batch(ctx, big), lzma_amd.decode_batch(ctx, small)
    kept_both = ctx.trim()
    assert kept_both >= 96 << 20                      # (at least the big call's output arena was being kept)
    assert lzma_amd.decode_batch(ctx, big) == want_big
    for _ in range(3):                                # three small calls in a row: the big call's blocks are let go
        assert lzma_amd.decode_batch(ctx, small) == want_small
    kept_small = ctx.trim()
    assert kept_small < 8 << 20 and ctx.trim() == 0
    assert lzma_amd.decode_batch(ctx, big) == want_big and all(g[1] == 0 for g in want_big)
