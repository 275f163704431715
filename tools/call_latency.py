"""Dev tool (GPU box): the time of ONE xlz_decode_batch call on small inputs -- host buffers in and out -- with its phases:
what a caller pays per call besides the decode itself (plan, pooled memory, launch, results, copies).
usage: python tools/call_latency.py"""
import sys, time
sys.path.insert(0, '.')
import corpus, lzma_amd
ctx = lzma_amd.Context(0)
for n, size in ((1, 4000), (64, 4000), (1, 1 << 20), (256, 65536)):
    ss = [lzma_amd.Stream(corpus.compress_alone(corpus.plain("T", 5 + i, size), preset=0), out_cap=size) for i in range(n)]
    lzma_amd.decode_batch(ctx, ss)
    t0 = time.perf_counter()
    for _ in range(30):
        lzma_amd.decode_batch(ctx, ss)
    dt = (time.perf_counter() - t0) / 30
    c = ctx.last_call_stats()
    print("%4d streams x %7d B: %.3f ms per call (upload %.2f decode %.2f download %.2f)" % (n, size, dt * 1e3, c["upload_ms"], c["decode_ms"], c["download_ms"]), flush=True)
