"""Dev tool (GPU box): ONE fed LZMA2 pull reader over a stream that resets its dictionary every MiB (the window
image of a session -- what a copy reads behind a dictionary reset, window.go:135-140 -- is brought up to date at every
reset; round 3 did that one byte per lane and allocated the image at NewReader2, round 4 copies 16 bytes per lane and
allocates it at the first reset that needs it).  Prints the reader's throughput; works with this tree's package and with
an older one (run it from that tree's root): the A/B is two runs.
    python tools/reader_resets.py [MiB of output] [dictionary MiB]"""
import io, os, sys, time
sys.path.insert(0, os.getcwd())
import corpus, lzma_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dict_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
enc = {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2}
segs = [corpus.plain("T", 600 + k, 1 << 20) for k in range(mib)]
blob = corpus.lzma2_concat(segs, dict_size=dict_mib << 20, preset=enc)
ctx = lzma_amd.Context(0)
best = None
for rep in range(3):
    r, err = lzma_amd.NewReader2(ctx, io.BytesIO(blob), dict_mib << 20, 1 << 20)   # fed: a session, not the unit-parallel refills
    assert err is None
    t0 = time.perf_counter()
    n, ok = 0, True
    while True:
        b, e = r.Read(1 << 20)
        ok = ok and b == segs[n >> 20][n & 0xFFFFF:(n & 0xFFFFF) + len(b)] if len(b) and (n & 0xFFFFF) + len(b) <= 1 << 20 else ok
        n += len(b)
        if e is not None:
            break
    dt = time.perf_counter() - t0
    assert n == mib << 20 and ok, (n, e)
    mem = r.memory() if hasattr(r, "memory") else None
    best = dt if best is None else min(best, dt)
print("fed NewReader2, %d MiB of text in 1 MiB segments that each reset a %d MiB dictionary: %.1f MiB/s (best of 3: %.2f s); "
      "refills %d, device memory (window, image) %s" % (mib, dict_mib, mib / best, best, r.stats()[0], mem))
