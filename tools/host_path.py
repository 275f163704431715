"""Dev tool (GPU box): host-buffers-in, host-buffers-out rate of xlz_decode_batch (the PCIe-inclusive path that a
drop-in caller sees) for several settings of the sliced form (xlz_ctx_set_slicing), with the call's phase times.
usage: python tools/host_path.py [family] [streams] [size] [distinct] [--slices 0,1,4,8] [--reps 4] [--preset6] [--lzma2 SEGMENTS] [--dict BYTES]   (--slices 0: the library's default rule)
The corpus is bench.py's (liblzma MODE_FAST / HC3) unless --preset6."""
import hashlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corpus, lzma_amd
from lzma_amd import _native as N

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = sys.argv[1:]
fam = args[0] if len(args) > 0 else "T"
n = int(args[1]) if len(args) > 1 else 4096
size = int(args[2]) if len(args) > 2 else 1 << 20
nd = int(args[3]) if len(args) > 3 else min(n, 512)
slices = [int(x) for x in opts[opts.index("--slices") + 1].split(",")] if "--slices" in opts else [1, 2, 4, 8]
reps = int(opts[opts.index("--reps") + 1]) if "--reps" in opts else 4
preset = 6 if "--preset6" in opts else {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2}
segments = int(opts[opts.index("--lzma2") + 1]) if "--lzma2" in opts else 0   # every stream: one raw LZMA2 stream of that many segments of `size` bytes
dict_size = int(opts[opts.index("--dict") + 1]) if "--dict" in opts else 65536
t0 = time.time()
if segments:
    cs, hs = corpus.make_lzma2_batch(fam, nd, segments, size, workers=min(os.cpu_count() or 1, 64), preset=preset, dict_size=dict_size)
    size *= segments
else:
    cs, hs = corpus.make_alone_batch(fam, nd, size, workers=min(os.cpu_count() or 1, 64), preset=preset, dict_size=dict_size)
print("corpus %d distinct x %d B in %.1f s, ratio %.3f" % (nd, size, time.time() - t0, sum(map(len, cs)) / (nd * size)), flush=True)
ctx = lzma_amd.Context(0)
print("library:", N.library_info(), flush=True)
ins = [np.frombuffer(cs[i % nd], dtype=np.uint8) for i in range(n)]
out = np.zeros((n, size), dtype=np.uint8)  # one host buffer per stream (rows), touched
descs = (N.StreamDesc * n)()
for i in range(n):
    descs[i].inp = ins[i].ctypes.data
    descs[i].in_len = ins[i].size
    descs[i].out = out[i].ctypes.data
    descs[i].out_cap = size
    descs[i].format = lzma_amd.FMT_LZMA2_RAW if segments else lzma_amd.FMT_LZMA_ALONE
    descs[i].dict_size = dict_size if segments else 0
res = (N.Result * n)()
for k in slices:
    if k == 0:
        ctx.set_slicing(0, 0, 0)   # the library's defaults
    else:
        ctx.set_slicing(1 if k > 1 else 0, max(1, n * size // max(k, 1)), k)
    times = []
    for rep in range(reps + 1):
        out[:] = 0
        t0 = time.perf_counter()
        st = N.lib().xlz_decode_batch(ctx._h, descs, n, res)
        dt = time.perf_counter() - t0
        assert st == 0, st
        c = ctx.last_call_stats()
        if rep:
            times.append(dt)
        print("  slices %2d run %d: %.1f ms = %.2f GiB/s | upload %.1f decode %.1f download %.1f ms, slices %d, sub-batches %d, "
              "slot occupancy %.3f" % (k, rep, dt * 1e3, n * size / dt / 2**30, c["upload_ms"], c["decode_ms"], c["download_ms"],
                                      c["slices"], c["sub_batches"], c["slot_occupancy"]), flush=True)
    bad = [i for i in range(n) if res[i].status != 0 or res[i].out_len != size]
    assert not bad, bad[:5]
    for i in list(range(0, n, max(1, n // 128))) + [n - 1]:
        assert hashlib.sha256(out[i].tobytes()).digest() == hs[i % nd], i
    med = statistics.median(times)
    print("slices %2d: median %.1f ms = %.2f GiB/s host to host (best %.2f), sample verified" % (
        k, med * 1e3, n * size / med / 2**30, n * size / min(times) / 2**30), flush=True)
