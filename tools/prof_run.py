"""Dev helper for rocprofv3: build one batch, run the decode kernel a few times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import corpus, lzma_amd
fam = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 16
nd = int(sys.argv[4]) if len(sys.argv) > 4 else 64
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 2
cs, hs = corpus.make_alone_batch(fam, nd, size, preset=6, workers=1)
ctx = lzma_amd.Context(0)
b = lzma_amd.Batch(ctx, [lzma_amd.Stream(cs[i % nd], out_cap=size) for i in range(n)])
for _ in range(runs):
    b.run(); b.sync()
print("kernel ms", b.kernel_ms())
