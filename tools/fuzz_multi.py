"""Differential fuzz of xlz_decode_batch_multi (dev tool, GPU box): LZMA2 streams long enough to be dealt to several
contexts in SLICES -- liblzma segments and packet-level crafted streams (every chunk kind and reset, rep matches that read
behind dictionary resets) concatenated into one stream, then damaged at random (bit flips, cuts, too little room) -- decoded
over three contexts on one device and compared with the CPU oracle on bytes, status and consumed input.  What it is after:
the fold of slice results and the whole-stream fallback (lzma_amd/csrc/xlz_host.hip: xlz_decode_batch_multi).
usage: python tools/fuzz_multi.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, lzma_amd, oracle
from lzma_amd import FMT_LZMA2_RAW, FMT_LZMA_ALONE, Stream
from lzma_craft import SMALL_PROPS, random_lzma2_stream

ENC = {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2}


def long_stream(rnd, ds):
    """-> a raw LZMA2 stream of many independent parts (each begins with a dictionary reset, or is glued on as it is)"""
    parts = []
    total = 0
    while total < rnd.choice([70_000, 150_000, 400_000]):
        k = rnd.random()
        if k < 0.55:
            p = corpus.plain(rnd.choice("TRMZ"), rnd.randrange(1 << 30), rnd.choice([500, 4000, 20_000, 60_000]))
            c = corpus.compress_raw_lzma2(p, dict_size=ds, preset=ENC)[:-1]
        else:
            c, _ = random_lzma2_stream(rnd, ds, max_chunks=6, max_packets=80, props=SMALL_PROPS)
            if c.endswith(b"\x00"):
                c = c[:-1]
        parts.append(c)
        total += len(c)
    return b"".join(parts) + b"\x00"


def damage(rnd, b):
    b = bytearray(b)
    k = rnd.random()
    if k < 0.35:
        return bytes(b)
    if k < 0.65:
        for _ in range(rnd.randint(1, 3)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        return bytes(b)
    if k < 0.8:
        return bytes(b[: rnd.randrange(len(b) // 2, len(b))])
    a = rnd.randrange(len(b) - 8)
    b[a:a + 4] = bytes(rnd.randrange(256) for _ in range(4))
    return bytes(b)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    ctxs = [lzma_amd.Context(0) for _ in range(3)]
    t_end = time.time() + budget
    n = n_bad = n_sliced = n_fallback_like = 0
    while time.time() < t_end:
        streams, wants = [], []
        for _ in range(rnd.randint(1, 6)):
            ds = rnd.choice([4096, 8192, 65536])
            blob = damage(rnd, long_stream(rnd, ds))
            full = oracle.lzma2_raw(blob, ds, 1 << 24)
            cap = len(full[0]) + rnd.choice([0, 0, 0, 100, -1, -5000])
            cap = max(cap, 0)
            streams.append(Stream(blob, FMT_LZMA2_RAW, out_cap=cap, dict_size=ds))
            wants.append(oracle.lzma2_raw(blob, ds, cap))
        for _ in range(rnd.randint(0, 8)):   # small LZMA1 streams beside them (dealt whole)
            p = corpus.plain("T", rnd.randrange(1 << 30), rnd.choice([300, 5000, 30_000]))
            c = corpus.compress_alone(p)
            streams.append(Stream(c, FMT_LZMA_ALONE, out_cap=len(p)))
            wants.append(oracle.lzma1_alone(c, len(p)))
        plan = lzma_amd.multi_plan(3, streams)
        n_sliced += sum(1 for it in plan if not it["whole"] and it["first"])
        got = lzma_amd.decode_batch_on(ctxs, streams)
        for i, (g, w) in enumerate(zip(got, wants)):
            if g != w:
                fn = os.path.join(ROOT, "gpurun_out", "fuzz_multi_fail_%d_%d.bin" % (seed, n + i))
                os.makedirs(os.path.dirname(fn), exist_ok=True)
                open(fn, "wb").write(streams[i].data)
                raise AssertionError("multi and oracle differ on stream %d (fmt %d, cap %d, dict %d): gpu (st %d, len %d, in %d) oracle "
                                     "(st %d, len %d, in %d) -> %s" % (i, streams[i].fmt, streams[i].out_cap, streams[i].dict_size, g[1],
                                                                        len(g[0]), g[2], w[1], len(w[0]), w[2], fn))
            n_bad += w[1] != 0
        n += len(streams)
        print("%d streams ok so far (%d with a non-OK status; %d dealt in slices)" % (n, n_bad, n_sliced), flush=True)
    print("multi fuzz ok: %d streams (%d with a non-OK status; %d dealt in slices)" % (n, n_bad, n_sliced))


if __name__ == "__main__":
    main()
