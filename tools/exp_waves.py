"""Experiment: literal-only synthetic streams (all-zero payload), throughput vs waves per CU."""
import sys, os, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lzma_amd
n, size = int(sys.argv[1]), 65536
blob = bytes([0x5D]) + struct.pack("<IQ", 65536, 0xFFFFFFFFFFFFFFFF) + bytes(8192)
ctx = lzma_amd.Context(0)
b = lzma_amd.Batch(ctx, [lzma_amd.Stream(blob, out_cap=size) for _ in range(n)])
for it in range(3):
    b.run(); b.sync()
ms = b.kernel_ms()
res = b.results()
print("per_cu", os.environ.get("XLZ_PER_CU"), "so", os.path.basename(os.environ.get("XLZ_SO", "default")),
      "%.2f ms -> %.2f GiB/s" % (ms, n * size / 2**30 / (ms / 1e3)), "status", res[0][1], "out", res[0][0],
      "zeros ok", b.download(0, 1000) == bytes(1000))
