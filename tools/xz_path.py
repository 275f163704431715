"""Dev tool (GPU box): a multi-block .xz file through xlz_xz_decode with the caller's buffers, for several block counts --
how the container path scales with the number of blocks (units) in the file.
    python tools/xz_path.py [blocks ...]      default: 256 1024 4096 (1 MiB of text-like data per block, CRC64)"""
import hashlib, os, sys, time
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

counts = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]
files = {}
with ProcessPoolExecutor(max_workers=min(os.cpu_count() or 1, 64)) as pool:   # before anything touches the GPU
    for n in counts:
        t0 = time.time()
        files[n] = bench.xz_file(pool, n, 1 << 20)
        print("corpus: %d blocks, %d bytes in %.1f s" % (n, len(files[n][0]), time.time() - t0), flush=True)
import lzma_amd
ctx = lzma_amd.Context(0)
for n in counts:
    data, want = files[n]
    _, total = lzma_amd.xz_index(data)
    out = np.zeros(total, dtype=np.uint8)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        got = lzma_amd.xz_decode_into(ctx, data, out, verify=True)
        dt = time.perf_counter() - t0
        cs = ctx.last_call_stats()
        if best is None or dt < best[0]:
            best = (dt, cs)
    assert got == total and hashlib.sha256(out).digest() == want
    dt, cs = best
    print("%5d blocks: %7.1f ms  %6.2f GiB/s host to host  (upload %.0f, decode %.0f, download %.0f, index + CRC64 %.0f ms; %d sub-batches, "
          "slot occupancy %.2f)" % (n, dt * 1e3, total / 2**30 / dt, cs["upload_ms"], cs["decode_ms"], cs["download_ms"],
                                    dt * 1e3 - cs["upload_ms"] - cs["decode_ms"] - cs["download_ms"], cs["sub_batches"], cs["slot_occupancy"]), flush=True)
