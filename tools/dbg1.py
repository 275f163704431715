import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import corpus, lzma_amd, oracle
from lzma_amd import Stream, FMT_LZMA2_RAW
ctx = lzma_amd.Context(0)
p = corpus.plain("T", 620, 150_000)
c = corpus.lzma2_concat([p[:50_000], p[50_000:100_000], p[100_000:]], dict_size=1 << 16)
for cap in (70_000, 50_000, 49_999, 50_001, 100_000, 120_000, 150_000, 149_999):
    g = lzma_amd.decode_batch(ctx, [Stream(c, FMT_LZMA2_RAW, out_cap=cap, dict_size=1 << 16)])[0]
    w = oracle.lzma2_raw(c, 1 << 16, cap)
    print(cap, "gpu", g[1], len(g[0]), g[2], "oracle", w[1], len(w[0]), w[2], "bytes equal", g[0] == w[0])
# single segment
c1 = corpus.compress_raw_lzma2(p[:50_000], dict_size=1 << 16)
for cap in (20_000, 49_999, 50_000):
    g = lzma_amd.decode_batch(ctx, [Stream(c1, FMT_LZMA2_RAW, out_cap=cap, dict_size=1 << 16)])[0]
    w = oracle.lzma2_raw(c1, 1 << 16, cap)
    print("single", cap, "gpu", g[1], len(g[0]), g[2], "oracle", w[1], len(w[0]), w[2], g[0] == w[0])
