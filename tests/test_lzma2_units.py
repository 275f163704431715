"""CPU: the unit plan of a raw LZMA2 stream (xlz_lzma2_units = the host scan scan_lzma2 behind every LZMA2 decode):
where a stream is cut into units that one wave each decodes concurrently.  A wrong cut is a silent data race on the
GPU (a unit reading window bytes another unit has not written yet), so the rule is pinned here without a GPU:
hand-computed plans for the shapes that matter, and two safety invariants on random chunk sequences --
  (1) history: an LZMA chunk that keeps the dictionary (control < 0xE0) sits in the same unit as every chunk back to
      the latest dictionary reset in front of it (Reader2.startChunk, reader2.go:100-173; window.go:135-140);
  (2) model: an LZMA chunk without new properties (control < 0xC0) sits in the same unit as the LZMA chunk before it
      (the state is carried across chunks, stored ones included, reader2.go:155-167).
The scan reads chunk HEADERS only, so the payloads here are arbitrary bytes."""
import random

import lzma_amd

UNIT = 256 << 10


def stored(n, reset):
    return bytes([1 if reset else 2]) + (n - 1).to_bytes(2, "big") + bytes(n)


def lzma(ctrl, unc, comp, props=0x5D):
    u = unc - 1
    h = bytes([ctrl | (u >> 16), (u >> 8) & 0xFF, u & 0xFF]) + (comp - 1).to_bytes(2, "big")
    if ctrl >= 0xC0:
        h += bytes([props])
    return h + bytes(comp)


def plan(chunks):
    blob = b"".join(chunks) + b"\x00"
    units = lzma_amd.lzma2_units(blob)
    # the units tile the input and the output
    assert units[0]["in_off"] == 0 and units[0]["out_off"] == 0
    for a, b in zip(units, units[1:]):
        assert a["in_off"] + a["in_len"] == b["in_off"] and a["out_off"] + a["out_len"] == b["out_off"]
    assert units[-1]["in_off"] + units[-1]["in_len"] == len(blob)
    return blob, units


def starts(chunks):
    pos, out = [], 0
    for c in chunks:
        pos.append(out)
        out += len(c)
    return pos


def test_hand_computed_plans():
    # ONE run of stored chunks, one dictionary reset (what xz writes for an incompressible file): a unit per 256 KiB
    _, u = plan([stored(65536, j == 0) for j in range(16)])
    assert [x["out_len"] for x in u] == [UNIT] * 4
    # liblzma's stored chunks hold a little less than 64 KiB: five of them until 256 KiB have gathered
    _, u = plan([stored(60_500, j == 0) for j in range(11)])
    assert [x["out_len"] for x in u] == [5 * 60_500, 5 * 60_500, 60_500]
    # segments that each reset the dictionary (cfg4-R): a unit per segment, no finer
    _, u = plan([stored(65536, j % 4 == 0) for j in range(16)])
    assert [x["out_len"] for x in u] == [UNIT] * 4
    # a run in front of an LZMA chunk that KEEPS the dictionary (0xC0: new properties, state reset): the chunk's matches
    # may read the stored bytes -- one unit
    _, u = plan([stored(65536, j == 0) for j in range(10)] + [lzma(0xC0, 5000, 2000)])
    assert len(u) == 1
    # ... in front of one that resets the dictionary (0xE0): the run is cut (256 KiB + 256 KiB + the rest), the chunk
    # starts a unit, the run behind it is a unit of its own (its first chunk resets the dictionary)
    _, u = plan([stored(65536, j == 0) for j in range(10)] + [lzma(0xE0, 360, 100)] + [stored(50_000, j == 0) for j in range(3)])
    assert [x["out_len"] for x in u] == [UNIT, UNIT, 2 * 65536, 360, 150_000]
    assert [x["have_reader"] for x in u] == [0, 0, 0, 0, 1]
    # a stored dictionary reset in front of an LZMA chunk WITHOUT new properties: the chunk continues the model of the
    # LZMA chunk in front of the stored one -- no cut
    _, u = plan([lzma(0xE0, 9000, 3000), stored(1000, True), lzma(0x80, 700, 300)])
    assert len(u) == 1
    # ... with new properties: the model starts over, the stored reset starts a unit
    _, u = plan([lzma(0xE0, 9000, 3000), stored(1000, True), lzma(0xC0, 700, 300)])
    assert [x["out_len"] for x in u] == [9000, 1700] and u[1]["have_reader"] == 1
    # independent segments (cfg4): a unit each
    _, u = plan([lzma(0xE0, 2000 + k, 900) for k in range(7)])
    assert len(u) == 7
    # stored chunks BEHIND an LZMA chunk without a reset of their own stay with it (nothing marks them as a run start)
    _, u = plan([lzma(0xE0, 9000, 3000)] + [stored(65536, False) for _ in range(12)])
    assert len(u) == 1
    # count only
    n = lzma_amd.ctypes.c_size_t()
    blob = b"".join(stored(65536, j == 0) for j in range(16)) + b"\x00"
    assert lzma_amd.N.lib().xlz_lzma2_units(blob, len(blob), None, 0, lzma_amd.ctypes.byref(n)) == 0 and n.value == 4
    one = (lzma_amd.N.Lzma2Unit * 1)()
    assert lzma_amd.N.lib().xlz_lzma2_units(blob, len(blob), one, 1, lzma_amd.ctypes.byref(n)) == lzma_amd.ERR_OUT_CAP


def test_no_cut_ever_separates_a_chunk_from_what_it_depends_on():
    rng = random.Random(20261004)
    for _ in range(3000):
        chunks, kinds = [], []   # kinds: ("S", reset) / ("L", ctrl)
        for _ in range(rng.randrange(1, 24)):
            r = rng.random()
            if r < 0.55:
                reset = rng.random() < 0.25
                n = rng.choice([1, 700, 30_000, 60_500, 65536])
                chunks.append(stored(n, reset))
                kinds.append(("S", reset))
            else:
                ctrl = rng.choice([0x80, 0x80, 0xA0, 0xC0, 0xE0, 0xE0])
                chunks.append(lzma(ctrl, rng.randrange(1, 200_000), rng.randrange(1, 40_000)))
                kinds.append(("L", ctrl))
        blob, units = plan(chunks)
        pos = starts(chunks)
        unit_of = []
        for p in pos:   # the unit each chunk starts in
            k = max(j for j, u in enumerate(units) if u["in_off"] <= p)
            unit_of.append(k)
        ustarts = {u["in_off"] for u in units}
        for p in ustarts:
            assert p in pos, "a unit starts inside a chunk"
        last_reset, last_lzma = 0, None   # chunk index of the latest dictionary reset / LZMA chunk
        for i, (k, v) in enumerate(kinds):
            if (k == "S" and v) or (k == "L" and v == 0xE0) or i == 0:
                last_reset = i
            if k == "L":
                if v < 0xE0:     # (1) keeps the dictionary: its history back to the latest reset is in its unit
                    assert unit_of[last_reset] == unit_of[i], (kinds, i)
                if v < 0xC0 and last_lzma is not None:   # (2) keeps the model of the LZMA chunk before it
                    assert unit_of[last_lzma] == unit_of[i], (kinds, i)
                last_lzma = i
            # a unit never starts at a stored chunk that keeps the dictionary unless its run is read by nothing:
            if i and pos[i] in ustarts and k == "S" and not v:
                j = i
                while j < len(kinds) and kinds[j] == ("S", False):
                    j += 1
                assert j == len(kinds) or kinds[j] == ("S", True) or kinds[j] == ("L", 0xE0), (kinds, i)


def test_damaged_headers_do_not_derail_the_scan():
    # a truncated header, properties >= 225, an end marker in the middle: the scan stops cutting, the units still tile the input
    for blob in (stored(65536, True)[:-100], stored(70, True) + b"\x01\x00", lzma(0xE0, 500, 100, props=230) + stored(5, True) + b"\x00",
                 stored(9, True) + b"\x00" + stored(9, True) + b"\x00", b"", b"\x00", b"\x7f"):
        units = lzma_amd.lzma2_units(blob)
        assert units[0]["in_off"] == 0 and sum(u["in_len"] for u in units) == len(blob)


def test_a_stream_beyond_4_gib_is_planned_with_64_bit_offsets():
    """the scan kept unit offsets in 32 bits, so the plan of a stream of 4 GiB and more wrapped silently.
    4.3 GiB of stored chunks (a sparse buffer: only the pages with chunk headers are touched): 65 600 chunks, one
    unit per 256 KiB, offsets beyond 2^32 intact"""
    import numpy as np
    n_chunks, body = 65_600, 65_536
    buf = np.zeros(n_chunks * (body + 3) + 1, dtype=np.uint8)
    at = np.arange(n_chunks, dtype=np.int64) * (body + 3)
    buf[at] = 2
    buf[0] = 1
    buf[at + 1] = 0xFF
    buf[at + 2] = 0xFF
    assert buf.size > 1 << 32
    units = lzma_amd.lzma2_units(buf)
    assert len(units) == n_chunks // 4
    assert units[0]["in_off"] == 0 and units[0]["out_off"] == 0
    for k in (1, 1000, len(units) - 1):
        assert units[k]["in_off"] == 4 * k * (body + 3) and units[k]["out_off"] == 4 * k * body
        assert units[k]["out_len"] == 4 * body or k == len(units) - 1
    last = units[-1]
    assert last["in_off"] > 1 << 32 and last["in_off"] + last["in_len"] == buf.size
    assert last["out_off"] + last["out_len"] == n_chunks * body


def test_batch_advice_counts_the_units_a_call_would_launch():
    """xlz_batch_advice: a C caller asks BEFORE uploading how much of the chip a call fills and whether
    the host's cores are the faster decoder -- units = streams for LZMA1, the plan of xlz_lzma2_units for LZMA2.  The rule
    compares estimated times: the host decodes ONE stream on ONE thread whatever its format (a Reader2 is one goroutine,
    reader2.go:216-250), 16 times as fast as a wave decodes a unit; for equal LZMA1 streams that is 16 units per host
    thread (bench.py's stream_count_sweep: 256 for 16 threads).  No GPU needed"""
    import corpus
    one = corpus.compress_alone(corpus.plain("T", 1, 5000), dict_size=1 << 16)
    a = lzma_amd.batch_advice([lzma_amd.Stream(one, out_cap=5000)] * 40, host_threads=16)
    assert a["units"] == 40 and a["break_even_units"] == 256 and a["prefer_cpu"] == 1
    assert a["wave_slots"] == 4096 and abs(a["fill"] - 40 / 4096) < 1e-9 and a["in_bytes"] == 40 * len(one)
    assert a["gpu_cost"] == len(one) and abs(a["cpu_cost"] - -(-40 * len(one) // 16) / 16) < 1e-9
    # 100 small LZMA1 streams on 16 threads: the host's
    assert lzma_amd.batch_advice([lzma_amd.Stream(one, out_cap=5000)] * 100, host_threads=16)["prefer_cpu"] == 1
    a = lzma_amd.batch_advice([lzma_amd.Stream(one, out_cap=5000)] * 300, host_threads=16)
    assert a["units"] == 300 and a["prefer_cpu"] == 0
    assert lzma_amd.batch_advice([lzma_amd.Stream(one, out_cap=5000)] * 300, host_threads=64)["prefer_cpu"] == 1
    # ONE LZMA2 stream of many dictionary-reset segments is many units: the GPU's job although it is one stream
    blob, units = plan([stored(65536, j % 2 == 0) for j in range(1200)])
    s2 = lzma_amd.Stream(blob, lzma_amd.FMT_LZMA2_RAW, out_cap=1200 * 65536, dict_size=1 << 16)
    a = lzma_amd.batch_advice([s2], host_threads=16)
    assert a["units"] == len(units) == 600 and a["prefer_cpu"] == 0
    both = lzma_amd.batch_advice([s2, lzma_amd.Stream(one, out_cap=5000)], host_threads=16)
    assert both["units"] == 601 and both["in_bytes"] == len(blob) + len(one)
    # ... and so is one of 100 units on a host of 16 threads ("100 units < 256" sent it to ONE host thread --
    # 100 units of serial work there, one unit's worth on the GPU), on any number of threads
    seg = corpus.compress_raw_lzma2(corpus.plain("T", 7, 20_000), dict_size=1 << 16)[:-1]
    hundred = seg * 100 + b"\0"
    sh = lzma_amd.Stream(hundred, lzma_amd.FMT_LZMA2_RAW, out_cap=100 * 20_000, dict_size=1 << 16)
    for threads in (1, 16, 256):
        a = lzma_amd.batch_advice([sh], host_threads=threads)
        assert a["units"] == 100 and a["prefer_cpu"] == 0, threads
        assert a["gpu_cost"] == len(seg) + 1 and abs(a["cpu_cost"] - len(hundred) / 16) < 1e-9   # (the last unit holds the end byte)
    # few big LZMA1 streams -- the shape of bench.py's cfg5-wrap, 64 streams -- are the host's: nothing inside LZMA1 is parallel
    big = corpus.compress_alone(corpus.plain("T", 9, 400_000), dict_size=1 << 23, lc=2, lp=1, pb=1)
    a = lzma_amd.batch_advice([lzma_amd.Stream(big, out_cap=400_000)] * 64, host_threads=16)
    assert a["units"] == 64 and a["prefer_cpu"] == 1
    # ... and one big stream among many small ones is still one thread's work on the host: 300 small streams + one that is
    # 20 times the others' sum
    a = lzma_amd.batch_advice([lzma_amd.Stream(one, out_cap=5000)] * 10 + [lzma_amd.Stream(big, out_cap=400_000)], host_threads=16)
    assert a["prefer_cpu"] == 1 and a["gpu_cost"] == len(big) and abs(a["cpu_cost"] - len(big) / 16) < 1e-9
    # one LZMA2 stream without inner resets is one unit -- one wave: decode it on the host
    single = corpus.compress_raw_lzma2(corpus.plain("T", 3, 300_000), dict_size=1 << 20)
    assert lzma_amd.lzma2_units(single)[0]["in_len"] == len(single)
    a = lzma_amd.batch_advice([lzma_amd.Stream(single, lzma_amd.FMT_LZMA2_RAW, out_cap=300_000, dict_size=1 << 20)], host_threads=1)
    assert a["units"] == 1 and a["prefer_cpu"] == 1 and a["break_even_units"] == 16
    empty = lzma_amd.batch_advice([], host_threads=4)
    assert empty["units"] == 0 and empty["fill"] == 0.0 and empty["prefer_cpu"] == 1


def _check_plan(items, streams, n_ctx):
    """every stream is covered exactly once by its items, in order; slices begin and end at unit boundaries"""
    by = {}
    for k, it in enumerate(items):
        by.setdefault(it["stream"], []).append(it)
        assert it["context"] < n_ctx
    assert sorted(by) == list(range(len(streams)))
    for i, its in by.items():
        s = streams[i]
        if its[0]["whole"]:
            assert len(its) == 1 and its[0]["in_off"] == 0 and its[0]["in_len"] == len(s.data)
            continue
        starts = {u["in_off"]: u for u in lzma_amd.lzma2_units(s.data)}
        assert its[0]["first"] and its[-1]["last"] and its[0]["in_off"] == 0 and its[0]["out_off"] == 0
        for a, b in zip(its, its[1:]):
            assert a["in_off"] + a["in_len"] == b["in_off"] and a["out_off"] + a["out_len"] == b["out_off"]
            assert not a["last"] and not b["first"]
        for it in its:
            assert it["in_off"] in starts and starts[it["in_off"]]["out_off"] == it["out_off"]
        assert its[-1]["in_off"] + its[-1]["in_len"] == len(s.data)


def test_the_units_of_one_stream_are_dealt_to_several_contexts():
    """SURVEY 8e, second clause: xlz_decode_batch_multi deals the UNITS of an LZMA2 stream that is a large
    part of the call to the contexts -- each gets a slice of the compressed input and a disjoint slice of the output,
    balanced by compressed bytes -- instead of landing the whole stream on one GPU.  The plan alone, without a GPU."""
    import corpus
    # ONE stream of 600 dictionary-reset units of unequal size, three contexts
    blob, units = plan([stored(2000 + 97 * (j % 50), True) for j in range(600)])
    big = lzma_amd.Stream(blob, lzma_amd.FMT_LZMA2_RAW, out_cap=sum(u["out_len"] for u in units), dict_size=1 << 16)
    items = lzma_amd.multi_plan(3, [big])
    _check_plan(items, [big], 3)
    assert len(items) == 12 and not items[0]["whole"]            # (4 pieces per context at most)
    load = [sum(it["in_len"] for it in items if it["context"] == c) for c in range(3)]
    assert max(load) - min(load) < 0.1 * sum(load) and min(load) > 0
    # a mixed call: the big stream, many small LZMA1 streams, one LZMA2 stream of a single unit
    one = corpus.compress_alone(corpus.plain("T", 1, 5000), dict_size=1 << 16)
    single = corpus.compress_raw_lzma2(corpus.plain("T", 3, 300_000), dict_size=1 << 20)
    streams = [lzma_amd.Stream(one, out_cap=5000) for _ in range(40)] + [big] + \
              [lzma_amd.Stream(single, lzma_amd.FMT_LZMA2_RAW, out_cap=300_000, dict_size=1 << 20)]
    items = lzma_amd.multi_plan(4, streams)
    _check_plan(items, streams, 4)
    assert sum(1 for it in items if it["stream"] == 40) > 4       # the big stream in slices
    assert [it["whole"] for it in items if it["stream"] == 41] == [True]
    load = [sum(it["in_len"] for it in items if it["context"] == c) for c in range(4)]
    assert max(load) < 1.25 * (sum(load) / 4)
    # too little room for what the headers announce: dealt whole (the single-GPU path settles status and bytes)
    tight = lzma_amd.Stream(blob, lzma_amd.FMT_LZMA2_RAW, out_cap=1000, dict_size=1 << 16)
    assert [it["whole"] for it in lzma_amd.multi_plan(3, [tight])] == [True]
    # a small stream is not worth cutting; one context gets everything whole
    small, _ = plan([stored(3000, True) for _ in range(10)])
    assert [it["whole"] for it in lzma_amd.multi_plan(3, [lzma_amd.Stream(small, lzma_amd.FMT_LZMA2_RAW, out_cap=30000)])] == [True]
    assert all(it["whole"] and it["context"] == 0 for it in lzma_amd.multi_plan(1, streams))


def test_how_a_call_is_cut_into_pieces():
    """xlz_decode_batch_plan (round 5, host only): the three forms of xlz_decode_batch.  Up to 6144 streams are ONE piece
    (one wave round of 16 / 20 / 24 per CU: slices); many short streams are a pipeline of overlapping pieces -- a quarter
    share, up to seven whole ones, a quarter share --; few rounds of LONG streams (256 KiB and more on average) are pieces of
    exactly one round of 4096 streams, one behind the other."""
    KiB, MiB = 1 << 10, 1 << 20
    assert lzma_amd.decode_batch_plan([MiB] * 4096) == ([0, 4096], 0)                 # BASELINE configs[1]
    assert lzma_amd.decode_batch_plan([MiB] * 6000) == ([0, 6000], 0)                 # one round of 24 per CU
    assert lzma_amd.decode_batch_plan([64 * KiB] * 7000) == ([0, 7000], 0)            # too small to bother
    assert lzma_amd.decode_batch_plan([]) == ([0, 0], 0)
    cuts, mode = lzma_amd.decode_batch_plan([64 * KiB] * 65536)                       # BASELINE configs[2]: the headline
    assert mode == 1 and len(cuts) == 10 and cuts[0] == 0 and cuts[-1] == 65536
    sizes = [b - a for a, b in zip(cuts, cuts[1:])]
    assert 2000 <= sizes[0] <= 2400 and 2000 <= sizes[-1] <= 2400 and all(8600 <= x <= 8900 for x in sizes[1:-1])
    assert lzma_amd.decode_batch_plan([2 * MiB] * 8192) == ([0, 4096, 8192], 2)       # BASELINE configs[4]
    assert lzma_amd.decode_batch_plan([MiB] * 16384) == ([0, 4096, 8192, 12288, 16384], 2)
    assert lzma_amd.decode_batch_plan([256 * KiB] * 12288) == ([0, 4096, 8192, 12288], 2)
    assert lzma_amd.decode_batch_plan([MiB] * 6200) == ([0, 4096, 6200], 2)
    cuts, mode = lzma_amd.decode_batch_plan([128 * KiB] * 16384)
    assert mode == 1 and len(cuts) == 6
    # unequal streams: the pieces are equal shares of the OUTPUT
    caps = [32 * KiB] * 30000 + [96 * KiB] * 30000
    cuts, mode = lzma_amd.decode_batch_plan(caps)
    assert mode == 1 and cuts[-1] == 60000
    share = [sum(caps[a:b]) for a, b in zip(cuts, cuts[1:])]
    assert max(share[1:-1]) < 1.1 * min(share[1:-1])
