"""Dev helper: small timing run of the decode kernel (not the official bench)."""
import sys, time, hashlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import corpus, lzma_amd
from lzma_amd import build
build.build()
fam = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 64   # distinct streams
preset = int(sys.argv[5]) if len(sys.argv) > 5 else 6
t = time.time()
cs, hs = corpus.make_alone_batch(fam, nd, size, preset=preset)
print("gen %.1fs ratio %.3f" % (time.time() - t, sum(map(len, cs)) / (nd * size)), flush=True)
ctx = lzma_amd.Context(0)
streams = [lzma_amd.Stream(cs[i % nd], out_cap=size) for i in range(n)]
b = lzma_amd.Batch(ctx, streams)
for it in range(3):
    b.run(); b.sync()
    ms = b.kernel_ms()
    print("run %d: %.2f ms  -> %.2f GiB/s out" % (it, ms, n * size / 2**30 / (ms / 1e3)), flush=True)
res = b.results()
bad = [i for i in range(n) if res[i][1] != 0 or res[i][0] != size]
print("bad", len(bad))
for i in range(0, n, max(1, n // 8)):
    assert hashlib.sha256(b.download(i, size)).digest() == hs[i % nd], i
print("verified")
