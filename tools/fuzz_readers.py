"""Differential fuzz of the PULL READERS (device sessions: pause / resume, moving windows, streaming input)
against the CPU oracle (dev tool, run on the GPU box).  Same stream generators as tools/fuzz_gpu.py;
every stream is read through NewReader1 / NewReader2 -- from bytes or from a file object in pieces --
with random Read sizes, and compared with the oracle on the bytes delivered and on how the reader ends
(io.EOF, or the error the oracle's status names; constructor errors included).
Two more modes take a share of the time: (b) `Reopen` -- a raw LZMA1 stream A cut at a random byte (clean io.EOF
inside a packet), then (*Reader1).Reopen on a stream B that continues A's model and window, against the oracle
on the LZMA2 stream E0(A cut) 80(B) that means the same (reader2.go:155-167); (c) eight readers at once on a
context with xlz_ctx_enable_batching (their refills share launches), each against the oracle.
usage: python tools/fuzz_readers.py [seconds] [seed]"""
import io
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import corpus
import lzma_amd
import oracle
from lzma_craft import ANY_PROPS, SMALL_PROPS, long_stale_lzma2_stream, random_lzma2_stream

CAP = 6 << 20  # the oracle needs an output bound; streams that would decode to more are skipped


def damage(rng, c, first=13):
    c = bytearray(c)
    k = int(rng.integers(0, 4))
    if k == 0 and len(c) > first + 8:
        return bytes(c[: int(rng.integers(first + 1, len(c)))])
    if k == 1 and len(c) > first + 8:
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(first, len(c)))] ^= 1 << int(rng.integers(0, 8))
        return bytes(c)
    if k == 2 and len(c) > first + 30:
        a = int(rng.integers(first, len(c) - 8))
        c[a:a + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        return bytes(c)
    return bytes(c)


def one_stream(rng):
    """-> (fmt, blob, dict_size)"""
    kind = rng.random()
    lc = int(rng.integers(0, 5)); lp = int(rng.integers(0, 5 - lc)); pb = int(rng.integers(0, 5))
    fam = "TRMZ"[int(rng.integers(0, 4))]
    n = int(rng.choice([1, 17, 300, 5000, 40000, 70000, 200000, 300001, 1 << 20, 3 << 20]))
    n = max(1, n + int(rng.integers(-3, 4)))
    p = corpus.plain(fam, int(rng.integers(1, 1 << 30)), n)
    if kind < 0.45:
        ds = int(rng.choice([4096, 4097, 65536, 100003, 1 << 20, 8 << 20]))
        c = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=bool(rng.random() < 0.3))
        if rng.random() < 0.4:
            c = damage(rng, c)
        return 1, c, 0
    if kind < 0.8:
        nseg = int(rng.integers(1, 5))
        cut = sorted(set(int(x) for x in rng.integers(0, n + 1, nseg - 1)))
        parts = [p[a:b] for a, b in zip([0] + cut, cut + [n]) if b > a] or [p]
        d2 = int(rng.choice([4096, 65536, 1 << 20, 8 << 20]))
        c = corpus.lzma2_concat(parts, dict_size=d2, lc=lc, lp=lp, pb=pb)
        if rng.random() < 0.4:
            c = damage(rng, c, first=0)
        return 2, c, d2
    r2 = random.Random(int(rng.integers(1, 1 << 62)))
    d2 = r2.choice([4096, 4097, 8192, 65536])
    if r2.random() < 0.04:  # a first dictionary epoch far longer than a session's sliding window, read behind later resets
        c, _ = long_stale_lzma2_stream(d2, first_epoch=r2.choice([200_000, 1_200_000, 2_300_000]), seed=r2.randrange(1 << 30))
    else:  # half of them renew the model with properties beyond lc+lp = 4 (reader2.go:159-165 takes any lc <= 8, lp <= 4)
        c, _ = random_lzma2_stream(r2, d2, props=ANY_PROPS if r2.random() < 0.5 else SMALL_PROPS)
    if rng.random() < 0.3 and len(c) > 8:
        c = bytearray(c)
        c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        c = bytes(c)
    return 2, c, d2


def reopen_case(ctx, rng):
    """-> None or a failure text"""
    import struct
    from lzma_craft import Encoder, Window, lzma2_lzma_chunk, props_byte
    r = random.Random(int(rng.integers(1, 1 << 62)))
    lc, lp, pb = r.choice([(3, 0, 2), (0, 0, 0), (1, 1, 1), (0, 2, 0), (4, 0, 0)])
    ds = r.choice([4096, 8192, 65536])
    w = Window(ds)
    e = Encoder(lc, lp, pb, ds, window=w)

    def packets(n):
        for _ in range(n):
            x = r.random()
            fill = w.size if w.full else w.pos
            if w.empty() or x < 0.4:
                e.literal(r.randrange(256) if r.random() < 0.3 else r.choice(b"abcde "))
            elif x < 0.6:
                e.match(r.randint(1, min(fill, ds)), r.choice([2, 3, 4, 8, 9, 17, 18, 40, 64, 65, 273]))
            elif x < 0.75:
                e.short_rep()
            else:
                e.rep(r.randrange(4), r.choice([2, 3, 9, 16, 17, 70, 273]))
    packets(r.randint(20, 400))
    pay_a, n_a = e.payload(), len(w.total)
    e.new_chunk()
    packets(r.randint(5, 200))
    pay_b, n_b = e.payload(), len(w.total) - n_a
    if n_b == 0 or len(pay_a) < 8:
        return None
    cut = r.choice([0, 0, 1, 2, 3, 5, 9, r.randrange(1, len(pay_a) - 5)])
    pa = pay_a[: max(6, len(pay_a) - cut)]
    framed = lzma2_lzma_chunk(0xE0, n_a, pa, props_byte(lc, lp, pb)) + lzma2_lzma_chunk(0x80, n_b, pay_b) + b"\x00"
    want, status, _ = oracle.lzma2_raw(framed, ds, n_a + n_b + 1024)
    rd, err = lzma_amd.NewLZMADecompressorForSevenZip(ctx, bytes([props_byte(lc, lp, pb)]) + struct.pack("<I", ds), n_a, [pa])
    if rd is None:
        return "constructor error %r" % getattr(err, "status", err)
    rd.__class__ = lzma_amd.Reader1
    out_a, e1 = rd.read_all(chunk=r.choice([1, 100, 4096, 70000]))
    if e1 is not None and cut:
        return None if status < 0 and out_a == want else "A (cut %d) ended with %r" % (cut, getattr(e1, "status", e1))
    if e1 is not None:
        return "A ended with %r" % getattr(e1, "status", e1)
    if out_a != want[: len(out_a)]:
        return "A's bytes differ"
    e0 = rd.Reopen(pay_b, n_b)
    if e0 is not None:
        return "Reopen returned %r" % getattr(e0, "status", e0)
    out_b, e2 = rd.read_all(chunk=r.choice([3, 500, 4096, 70000]))
    if out_a + out_b != want:
        return "after Reopen: reader %d + %d bytes, oracle %d (cut %d)" % (len(out_a), len(out_b), len(want), cut)
    if (e2 is None) != (status >= 0):
        return "after Reopen: ended with %r, oracle status %d" % (getattr(e2, "status", e2), status)
    return None


def concurrent_case(bctx, rng):
    """eight readers on the batching context at once -> None or a failure text"""
    import threading
    jobs = []
    while len(jobs) < 8:
        fmt, c, ds = one_stream(rng)
        want = oracle.lzma1_alone(c, CAP) if fmt == 1 else oracle.lzma2_raw(c, ds, CAP)
        if want[1] != oracle.ERR_OUT_CAP:
            jobs.append((fmt, c, ds, want, int(rng.choice([100, 4096, 65536, 1 << 20]))))
    res = [None] * len(jobs)

    def work(i):
        fmt, c, ds, want, chunk = jobs[i]
        try:
            rd, err = lzma_amd.NewReader1(bctx, c) if fmt == 1 else lzma_amd.NewReader2(bctx, c, ds)
            if rd is None:
                res[i] = None if (want[1] < 0 and err.status == want[1] and not want[0]) else "constructor error %d" % err.status
                return
            out, e = rd.read_all(chunk=chunk)
            if out != want[0]:
                res[i] = "bytes differ: reader %d, oracle %d" % (len(out), len(want[0]))
            elif (want[1] < 0) != isinstance(e, lzma_amd.LzmaError) or (want[1] < 0 and e.status != want[1]):
                res[i] = "ended with %r, oracle status %d" % (getattr(e, "status", e), want[1])
        except Exception as ex:  # noqa: BLE001
            res[i] = "exception %r" % ex
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i, f in enumerate(res):
        if f:
            return "reader %d of 8 (fmt %d, dict %d, len %d): %s" % (i, jobs[i][0], jobs[i][2], len(jobs[i][1]), f), jobs[i][1]
    return None


def fuzz(ctx, budget, seed, verbose=True):
    """-> (readers compared, readers that ended in an error); raises AssertionError on a mismatch"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    n_total = n_err = n_skipped = 0
    n_unsup = [0]
    n_reopen = n_conc = 0
    bctx = None
    while time.time() < t_end:
        mode = rng.random()
        if mode < 0.25:
            f = reopen_case(ctx, rng)
            n_total += 1
            n_reopen += 1
            if f:
                print("MISMATCH reopen case: %s" % f, flush=True)
                raise AssertionError("reader and oracle differ (Reopen: %s)" % f)
            continue
        if mode < 0.32:
            if bctx is None:
                bctx = lzma_amd.Context(0)
                bctx.enable_batching(window_us=2000, max_streams=64)
            f = concurrent_case(bctx, rng)
            n_total += 8
            n_conc += 8
            if f:
                os.makedirs("gpurun_out", exist_ok=True)
                fn = "gpurun_out/fuzz_reader_fail_%d_%d.bin" % (seed, n_total)
                open(fn, "wb").write(f[1])
                print("MISMATCH concurrent readers: %s -> %s" % (f[0], fn), flush=True)
                raise AssertionError("reader and oracle differ (%s), input saved as %s" % (f[0], fn))
            continue
        fmt, c, ds = one_stream(rng)
        want = oracle.lzma1_alone(c, CAP) if fmt == 1 else oracle.lzma2_raw(c, ds, CAP)
        if want[1] == oracle.ERR_OUT_CAP:
            n_skipped += 1
            continue
        piece = int(rng.choice([64, 1000, 4096, 65536, 1 << 20]))
        streaming = bool(rng.random() < 0.5) and len(c) > piece
        src = io.BytesIO(c) if streaming else c
        r, err = (lzma_amd.NewReader1(ctx, src, piece) if fmt == 1 else lzma_amd.NewReader2(ctx, src, ds, piece))
        what = "fmt %d dict %d len %d piece %s" % (fmt, ds, len(c), piece if streaming else "-")
        fail = None
        if r is None:
            if not (want[1] < 0 and err.status == want[1] and len(want[0]) == 0):
                fail = "constructor error %d, oracle (st %d, len %d)" % (err.status, want[1], len(want[0]))
        else:
            out, e = [], None
            sizes = [int(x) for x in rng.choice([1, 3, 100, 4096, 5000, 65536, 100000, 1 << 20], 6)]
            k = got = 0
            while True:
                b, e = r.Read(sizes[k % len(sizes)])
                k += 1
                out.append(b)
                got += len(b)
                if e is not None:
                    break
                assert got <= CAP + (1 << 20), "reader runs past the oracle's output"
            out = b"".join(out)
            # (round 2 exempted fed readers that ended in XLZ_ERR_UNSUPPORTED -- reads behind a dictionary reset, models
            #  beyond lc+lp = 4 -- as a documented limit.  The limit is gone: sessions keep the window image and grow
            #  their model, so a fed reader must equal the oracle like any other.)
            if fail:
                pass
            elif out != want[0]:
                d = next((i for i in range(min(len(out), len(want[0]))) if out[i] != want[0][i]), None)
                fail = "bytes differ: reader %d, oracle %d, first diff %s" % (len(out), len(want[0]), d)
            elif want[1] < 0:
                if not (isinstance(e, lzma_amd.LzmaError) and e.status == want[1]):
                    fail = "reader ended with %r, oracle status %d" % (getattr(e, "status", e), want[1])
            elif e is not lzma_amd.io_EOF:
                fail = "reader ended with %r, oracle status %d" % (getattr(e, "status", e), want[1])
        n_total += 1
        n_err += want[1] < 0
        if fail:
            fn = "gpurun_out/fuzz_reader_fail_%d_%d.bin" % (seed, n_total)
            os.makedirs("gpurun_out", exist_ok=True)
            open(fn, "wb").write(c)
            print("MISMATCH %s: %s -> %s" % (what, fail, fn), flush=True)
            raise AssertionError("reader and oracle differ (%s), input saved as %s" % (fail, fn))
        if verbose and n_total % 50 == 0:
            print("%d readers ok so far (%d ending in an error, %d skipped, %d Reopen cases, %d concurrent)"
                  % (n_total, n_err, n_skipped, n_reopen, n_conc), flush=True)
    if bctx is not None:
        bctx.close()
    return n_total, n_err


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n, ne = fuzz(lzma_amd.Context(0), budget, seed)
    print("reader fuzz ok: %d readers (%d ending in an error)" % (n, ne))
