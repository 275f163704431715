"""Dev tool: decode a few small streams on the GPU and show where they first differ from the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import corpus, lzma_amd, oracle
from lzma_amd import Stream, FMT_LZMA_ALONE
ctx = lzma_amd.Context(0)
for fam, n in (("R", 4000), ("T", 4000), ("Z", 4000), ("T", 100000)):
    p = corpus.plain(fam, 7, n)
    c = corpus.compress_alone(p)
    g = lzma_amd.decode_batch(ctx, [Stream(c, FMT_LZMA_ALONE, out_cap=n)])[0]
    w = oracle.lzma1_alone(c, n)
    d = next((i for i in range(min(len(g[0]), len(w[0]))) if g[0][i] != w[0][i]), None)
    print(fam, n, "gpu status/len/consumed", g[1], len(g[0]), g[2], "oracle", w[1], len(w[0]), w[2], "first diff", d)
    if d is not None:
        print("  gpu ", g[0][max(0, d - 8):d + 8].hex(), "\n  want", w[0][max(0, d - 8):d + 8].hex())
