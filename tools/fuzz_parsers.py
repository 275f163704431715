"""Mutation fuzz of the HOST-ONLY parsers (xlz_xz_index, xlz_7z_index without an encoded header, xlz_lzma2_units -- the
chunk-header scan behind every raw LZMA2 decode) and planners (xlz_batch_advice, xlz_decode_batch_plan,
xlz_decode_batch_multi_plan): no GPU needed.  Valid .xz files (liblzma) and hand-built .7z archives (tests/sevenzip_craft.py) are truncated,
bit-flipped, overwritten near their ends / inside the 7z end header (its CRCs fixed up so that the parser
gets that far).  A parse must either fail with a status or describe byte ranges inside the file.
Under AddressSanitizer (CPU build only; the pool refuses GPU ASan):
    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fno-gpu-rdc -fsanitize=address -fno-gpu-sanitize \\
          -shared-libasan -I include lzma_amd/csrc/*.hip -o build_ab/libxlz_asan.so
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=<clang rt dir>/libclang_rt.asan-x86_64.so XLZ_SO=$PWD/build_ab/libxlz_asan.so \\
          python tools/fuzz_parsers.py 60 1
(round 2: 1 267 083 .xz and 579 109 .7z inputs, no report.)
usage: python tools/fuzz_parsers.py [seconds per parser] [seed]"""
import lzma
import os
import random
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lzma_amd
from sevenzip_craft import archive, bcj_lzma_folder, copy_folder, lzma2_folder, lzma_folder


def _mutate(rnd, b, tail_from=None):
    b = bytearray(b)
    k = rnd.randrange(6)
    if k == 0 and len(b) > 2:
        return b[:rnd.randrange(1, len(b))]
    if k == 1:
        for _ in range(rnd.randrange(1, 6)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    lo = tail_from if tail_from is not None else max(0, len(b) - 64)
    span = max(1, len(b) - lo)
    if k in (2, 5):
        for _ in range(rnd.randrange(1, 8)):
            b[lo + rnd.randrange(span)] = rnd.randrange(256) if rnd.random() < 0.7 else rnd.choice([0, 0xFF, 0x80, 0xE0])
    if k == 3:
        i = rnd.randrange(len(b))
        b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
    if k == 4:
        i = lo + rnd.randrange(span)
        n = rnd.randrange(1, 9)
        b[i:i + n] = b"\xff" * n
    return b


def fuzz_xz(seconds, seed):
    """-> (inputs, inputs that parsed)"""
    rnd = random.Random(seed)
    xs = []
    for n in (0, 1, 100, 5000, 70000):
        p = bytes(rnd.randrange(97, 105) for _ in range(n))
        for chk in (lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256):
            xs.append(lzma.compress(p, format=lzma.FORMAT_XZ, check=chk, preset=0))
    xs.append(xs[3] + xs[7])                # two streams
    xs.append(xs[5] + b"\0" * 8 + xs[9])    # stream padding
    t_end = time.time() + seconds
    n = ok = 0
    while time.time() < t_end:
        b = bytes(_mutate(rnd, rnd.choice(xs)))
        n += 1
        try:
            blocks, total = lzma_amd.xz_index(b)
        except lzma_amd.LzmaError:
            continue
        ok += 1
        for bl in blocks:
            assert bl["comp_off"] + bl["comp_len"] <= len(b), (bl, len(b))
    return n, ok


def fuzz_7z(seconds, seed):
    """-> (inputs, inputs that parsed)"""
    rnd = random.Random(seed)
    files = [bytes(rnd.randrange(97, 110) for _ in range(n)) for n in (1, 50, 3000, 20000)]
    makers = [lambda d: lzma_folder(d), lambda d: lzma2_folder(d), lambda d: copy_folder(d),
              lambda d: lzma_folder(d, dict_size=1 << 20, lc=0, lp=2, pb=0)]
    samples = []
    for combo in ([0], [0, 1], [1, 2, 3], [3, 3, 0]):
        fo = []
        for c in combo:
            parts = files[: 1 + (c % 3)]
            rec, packed = makers[c](b"".join(parts))
            fo.append((rec, packed, parts))
        for ws in (True, False):
            for fc in (True, False):
                a = archive(fo, with_substreams=ws, folder_crc=fc)
                samples.append((a, 32 + struct.unpack("<Q", a[12:20])[0]))
    # folders with complex coders (flag 0x10: explicit stream counts) -- round 2's fuzz never produced n_in = 0, which
    # crashed place_folders; half of the mutations of these samples rewrite a count byte behind a 0x1x / 0x3x
    # coder byte to 0 / 1 / 2 / 33
    complex_samples = []
    for data in (files[1], files[2]):
        rec, packed = bcj_lzma_folder(data)
        for ws in (True, False):
            a = archive([(rec, packed, [data])], with_substreams=ws, folder_crc=True)
            complex_samples.append((a, 32 + struct.unpack("<Q", a[12:20])[0]))
        crec, cpacked = lzma_folder(data)
        raw = ("raw", b"\x01" + bytes([crec[0] | 0x10]) + crec[1:4] + b"\x01\x01" + crec[4:])   # one coder, counts spelled out
        a = archive([(raw, cpacked, [data])], folder_crc=True)
        complex_samples.append((a, 32 + struct.unpack("<Q", a[12:20])[0]))
    t_end = time.time() + seconds
    n = ok = 0
    while time.time() < t_end:
        if rnd.random() < 0.25:
            a, hs = rnd.choice(complex_samples)
            b = bytearray(a)
            if rnd.random() < 0.5:
                cand = [i for i in range(hs, len(b) - 6) if b[i] & 0xD0 == 0x10 and (b[i] & 0x0F) in (1, 3, 4)]
                if cand:
                    i = rnd.choice(cand)
                    b[i + 1 + (b[i] & 0x0F) + rnd.randrange(2)] = rnd.choice([0, 0, 1, 2, 33])
            else:
                b = _mutate(rnd, a, tail_from=hs)
        else:
            a, hs = rnd.choice(samples)
            b = _mutate(rnd, a, tail_from=hs)
        if len(b) > max(hs, 32) and rnd.random() < 0.8:  # the CRCs of the end header and of the start header
            b[28:32] = struct.pack("<I", zlib.crc32(bytes(b[hs:])) & 0xFFFFFFFF)
            b[8:12] = struct.pack("<I", zlib.crc32(bytes(b[12:32])) & 0xFFFFFFFF)
        b = bytes(b)
        n += 1
        try:
            folders, subs, total = lzma_amd.sevenzip_index(b)
        except lzma_amd.LzmaError:
            continue
        ok += 1
        for f in folders:
            assert f["pack_off"] + f["pack_len"] <= len(b), (f, len(b))
    return n, ok


def fuzz_lzma2_units(seconds, seed):
    """-> (inputs, units planned).  Raw LZMA2 streams written by liblzma (text: LZMA chunks; random data: stored chunks;
    concatenations with dictionary resets) and random chunk-header sequences, mutated: the plan's units must tile the
    input exactly, whatever the bytes say."""
    import corpus
    rnd = random.Random(seed)
    xs = []
    for fam, n in (("T", 300_000), ("R", 200_000), ("M", 400_000), ("Z", 100_000)):
        xs.append(corpus.compress_raw_lzma2(corpus.plain(fam, seed + n, n), dict_size=1 << 16, preset=0))
    xs.append(corpus.lzma2_concat([corpus.plain("TR"[i % 2], seed + i, 70_000) for i in range(6)], dict_size=1 << 16, preset=0))
    t_end = time.time() + seconds
    n = units = 0
    while time.time() < t_end:
        if rnd.random() < 0.5:
            b = bytes(_mutate(rnd, rnd.choice(xs), tail_from=0))
        else:   # headers only: control bytes of every kind with random sizes, bodies mostly missing
            b = bytearray()
            for _ in range(rnd.randrange(1, 12)):
                c = rnd.choice([0, 1, 2, 3, 0x7F, 0x80, 0x9F, 0xA0, 0xC0, 0xE0, 0xFF, rnd.randrange(256)])
                b += bytes([c]) + bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 6)))
                if rnd.random() < 0.3:
                    b += bytes(rnd.randrange(1, 70000))
            b = bytes(b)
        n += 1
        plan = lzma_amd.lzma2_units(b)
        assert plan[0]["in_off"] == 0 and sum(u["in_len"] for u in plan) == len(b), (len(b), plan[:3])
        for u, v in zip(plan, plan[1:]):
            assert u["in_off"] + u["in_len"] == v["in_off"] and u["out_off"] + u["out_len"] == v["out_off"]
        units += len(plan)
    return n, units


def fuzz_plans(seconds, seed):
    """-> (calls, items dealt).  The host-only planners behind a batch call: xlz_batch_advice (the break-even rule),
    xlz_decode_batch_plan (how a call is cut into pieces) and xlz_decode_batch_multi_plan (what goes to which GPU) on
    descriptors of every format with mutated inputs, odd capacities and odd properties.  No device, nothing is decoded: the
    answers must be consistent with the descriptors whatever the bytes say."""
    import corpus
    rnd = random.Random(seed)
    alone = [corpus.compress_alone(corpus.plain(f, seed + n, n), dict_size=1 << 16, known_size=k, preset=0)
             for f, n, k in (("T", 20_000, True), ("R", 5_000, True), ("M", 60_000, False), ("Z", 100, True))]
    raw2 = [corpus.compress_raw_lzma2(corpus.plain(f, seed + n, n), dict_size=1 << 16, preset=0) for f, n in (("T", 300_000), ("R", 150_000))]
    raw2.append(corpus.lzma2_concat([corpus.plain("TR"[i % 2], seed + i, 70_000) for i in range(40)], dict_size=1 << 16, preset=0))
    t_end = time.time() + seconds
    calls = dealt = 0
    while time.time() < t_end:
        streams = []
        for _ in range(rnd.choice([0, 1, 1, 2, 5, 40, 300])):
            k = rnd.randrange(4)
            cap = rnd.choice([0, 1, 4096, 70_000, 300_000, 1 << 22, (1 << 32) + 5, rnd.randrange(1 << 20)])
            if k == 0:
                b = rnd.choice(alone)
                b = bytes(_mutate(rnd, b, tail_from=0)) if rnd.random() < 0.5 else b
                streams.append(lzma_amd.Stream(b, lzma_amd.FMT_LZMA_ALONE, out_cap=cap))
            elif k == 1:
                b = rnd.choice(raw2)
                b = bytes(_mutate(rnd, b, tail_from=0)) if rnd.random() < 0.5 else b
                streams.append(lzma_amd.Stream(b, lzma_amd.FMT_LZMA2_RAW, out_cap=cap, dict_size=rnd.choice([0, 4096, 1 << 16, 1 << 26, 0xFFFFFFFF])))
            elif k == 2:
                b = rnd.choice(alone)[13:]
                streams.append(lzma_amd.Stream(b, lzma_amd.FMT_LZMA_RAW, out_cap=cap, dict_size=rnd.choice([0, 1 << 16, 0xFFFFFFFF]),
                                               unpack_size=rnd.choice([cap, lzma_amd.UNKNOWN_SIZE, 0]), props=rnd.randrange(256)))
            else:
                streams.append(lzma_amd.Stream(bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 40))),
                                               rnd.choice([lzma_amd.FMT_LZMA_ALONE, lzma_amd.FMT_LZMA2_RAW, lzma_amd.FMT_LZMA_RAW, 7]), out_cap=cap,
                                               props=rnd.randrange(256)))
        calls += 1
        threads = rnd.choice([0, 1, 16, 192, 1 << 20])
        try:
            adv = lzma_amd.batch_advice(streams, host_threads=threads)
            assert adv["prefer_cpu"] in (0, 1) and adv["in_bytes"] == sum(len(s.data) for s in streams), adv
            assert adv["cpu_cost"] >= 0 and adv["gpu_cost"] >= 0 and 0 <= adv["fill"], adv   # (a stream that does not plan launches no unit)
        except lzma_amd.LzmaError:
            pass
        cuts, mode = lzma_amd.decode_batch_plan([s.out_cap for s in streams])
        assert mode in (0, 1, 2) and cuts[0] == 0 and cuts[-1] == len(streams) and all(a < b for a, b in zip(cuts, cuts[1:])) or not streams, (cuts, mode)
        n_ctx = rnd.choice([1, 2, 3, 8])
        try:
            items = lzma_amd.multi_plan(n_ctx, streams)
        except lzma_amd.LzmaError:
            continue
        dealt += len(items)
        seen = {}
        for it in items:   # every stream's items: adjacent, in order, inside the descriptor, on a context that exists
            s = streams[it["stream"]]
            assert 0 <= it["context"] < n_ctx and it["in_off"] + it["in_len"] <= len(s.data) and it["out_off"] + it["out_len"] <= max(s.out_cap, it["out_off"] + it["out_len"] if it["whole"] else 0), (it, len(s.data), s.out_cap)
            at = seen.get(it["stream"])
            assert (at is None) == it["first"] and (at is None or at == (it["in_off"], it["out_off"])), (it, at)
            seen[it["stream"]] = (it["in_off"] + it["in_len"], it["out_off"] + it["out_len"])
        assert sorted(seen) == list(range(len(streams))), (sorted(seen)[:5], len(streams))
    return calls, dealt


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("xz_index: %d inputs, %d parsed" % fuzz_xz(secs, seed))
    print("7z_index: %d inputs, %d parsed" % fuzz_7z(secs, seed))
    print("lzma2_units: %d inputs, %d units planned" % fuzz_lzma2_units(secs, seed))
    print("advice / piece plan / multi-GPU plan: %d calls, %d items dealt" % fuzz_plans(secs, seed))
