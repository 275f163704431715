"""Dev tool: host-buffers-in, host-buffers-out rate of xlz_decode_batch (the PCIe-inclusive path that a
drop-in caller sees), next to the device-resident kernel rate.
usage: python tools/host_path.py [family] [streams] [size] [distinct]"""
import ctypes, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corpus, lzma_amd
from lzma_amd import _native as N

fam = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
size = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
nd = int(sys.argv[4]) if len(sys.argv) > 4 else min(n, 256)
t0 = time.time()
cs, hs = corpus.make_alone_batch(fam, nd, size, workers=min(os.cpu_count() or 1, 64))
print("corpus %d distinct x %d B in %.1f s" % (nd, size, time.time() - t0), flush=True)
ctx = lzma_amd.Context(0)
ins = [np.frombuffer(cs[i % nd], dtype=np.uint8) for i in range(n)]
out = np.zeros((n, size), dtype=np.uint8)  # one host buffer per stream (rows), touched
descs = (N.StreamDesc * n)()
for i in range(n):
    descs[i].inp = ins[i].ctypes.data
    descs[i].in_len = ins[i].size
    descs[i].out = out[i].ctypes.data
    descs[i].out_cap = size
    descs[i].format = lzma_amd.FMT_LZMA_ALONE
res = (N.Result * n)()
for rep in range(4):
    out[:] = 0
    t0 = time.perf_counter()
    st = N.lib().xlz_decode_batch(ctx._h, descs, n, res)
    dt = time.perf_counter() - t0
    assert st == 0, st
    print("xlz_decode_batch run %d: %.1f ms -> %.2f GiB/s host to host" % (rep, dt * 1e3, n * size / dt / 2**30), flush=True)
bad = [i for i in range(n) if res[i].status != 0 or res[i].out_len != size]
assert not bad, bad[:5]
for i in list(range(0, n, max(1, n // 64))) + [n - 1]:
    assert hashlib.sha256(out[i].tobytes()).digest() == hs[i % nd], i
print("verified sample bit-exact")
