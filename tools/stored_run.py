"""Dev tool (GPU box): ONE raw LZMA2 stream of stored chunks with a single dictionary reset -- what xz writes for one
incompressible file -- through the device-resident batch path: units, kernel time, GiB/s.
    python tools/stored_run.py [MiB]         (XLZ_STORED_UNIT_KIB=0: runs of stored chunks are cut at dictionary resets only)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import corpus, lzma_amd
from lzma_amd import Stream, FMT_LZMA2_RAW

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = corpus.plain("R", 4242, mib << 20)
t0 = time.time()
blob = corpus.compress_raw_lzma2(p, dict_size=1 << 16, preset=0)
assert blob[0] == 1  # (the one dictionary reset, at the start; every later chunk is 0x02)
ctx = lzma_amd.Context(0)
b = lzma_amd.Batch(ctx, [Stream(blob, FMT_LZMA2_RAW, out_cap=len(p), dict_size=1 << 16)])
b.run()
b.sync()
ctx.event_record(0)
steps = 3
for _ in range(steps):
    b.run()
ctx.event_record(1)
b.sync()
ms = ctx.event_elapsed_ms(0, 1) / steps
res = b.results()
assert res[0][0] == len(p) and res[0][1] == 0, res
assert hashlib.sha256(b.download(0, len(p))).digest() == hashlib.sha256(p).digest()
print("XLZ_STORED_UNIT_KIB=%s: %d MiB in one run of stored chunks: %d units, %.3f ms per launch, %.1f GiB/s decoded, bit-exact"
      % (os.environ.get("XLZ_STORED_UNIT_KIB", "(default)"), mib, b.stats()[2], ms, mib / 1024 / (ms / 1e3)))
