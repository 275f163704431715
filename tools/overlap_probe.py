"""Dev tool (GPU box): does a host<->device copy on a SECOND stream make progress while the decode launch runs?
Round 2 built sub-batch overlap three ways and saw the D2H copy of sub-batch k finish together with the decode of
sub-batch k+1.  This probe separates the possible causes:

  * copy engine: SDMA or a blit kernel (`__amd_rocclr_copyBuffer`) that needs wave slots next to the persistent grid
    -- run under `rocprofv3 --kernel-trace --memory-copy-trace` to see which; HSA_ENABLE_SDMA=0 forces blits;
  * wave priority: the decode waves raise their own priority (s_setprio 1..3); a blit wave at priority 0 on the same
    SIMD would starve (XLZ_SO=<build with -DXLZ_NO_SETPRIO> to test);
  * occupancy: grid of 16 waves per CU leaves wave slots free, LDS too (7.9 KB x 16 of 160 KB).

It times, with events on the copy stream: D2H of `mib` MiB from the batch's output arena into pinned memory alone, the
decode alone, then both at once (copy enqueued right after the launch), and H2D likewise.
usage: python tools/overlap_probe.py [streams] [copy MiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import corpus
import lzma_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
size = 1 << 20
nd = 128
cs, hs = corpus.make_alone_batch("T", nd, size, workers=min(os.cpu_count() or 1, 32))
torch.cuda.set_device(0)
ctx = lzma_amd.Context(0)
batch = lzma_amd.Batch(ctx, [lzma_amd.Stream(cs[i % nd], out_cap=size) for i in range(n)])
batch.run()
batch.sync()
print("library:", lzma_amd._native.library_info(), flush=True)

nbytes = mib << 20
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
dev.fill_(7)
host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
s2 = torch.cuda.Stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def copy(direction):
    with torch.cuda.stream(s2):
        e0.record(s2)
        if direction == "d2h":
            host.copy_(dev, non_blocking=True)
        else:
            dev.copy_(host, non_blocking=True)
        e1.record(s2)


def decode_ms():
    t0 = time.perf_counter()
    batch.run()
    batch.sync()
    return (time.perf_counter() - t0) * 1e3, batch.kernel_ms()


for direction in ("d2h", "h2d"):
    copy(direction)
    s2.synchronize()
    copy(direction)
    s2.synchronize()
    alone = e0.elapsed_time(e1)
    print("%s alone: %.1f ms for %d MiB = %.1f GB/s" % (direction, alone, mib, nbytes / alone / 1e6), flush=True)
    wall, k = decode_ms()
    print("decode alone: wall %.1f ms, kernel %.1f ms" % (wall, k), flush=True)
    # both: launch the decode, then enqueue the copy on the other stream at once
    t0 = time.perf_counter()
    batch.run()
    copy(direction)
    s2.synchronize()
    t_copy_done = (time.perf_counter() - t0) * 1e3
    batch.sync()
    t_all = (time.perf_counter() - t0) * 1e3
    print("%s under the decode launch: copy events %.1f ms (alone %.1f), copy done %.1f ms after the launch, decode "
          "done at %.1f ms (kernel %.1f ms)" % (direction, e0.elapsed_time(e1), alone, t_copy_done, t_all, batch.kernel_ms()),
          flush=True)
res = batch.results()
assert all(r[1] == 0 and r[0] == size for r in res)
print("decode results ok")
