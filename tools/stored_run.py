"""Dev tool (GPU box): raw LZMA2 streams of stored chunks with a single dictionary reset each -- what xz writes for one
incompressible file -- through the device-resident batch path: units, kernel time, GiB/s, fraction of the HBM roof, slot
occupancy, and beside it what a plain device-to-device hipMemcpy of the same bytes reaches on the same box (the copy
ceiling).  Sizes in MiB; a size above 3584 is made of several streams (a stream holds less than 4 GiB).  The streams are
written directly (chunk headers + random bytes: liblzma needs minutes for gigabytes of incompressible data).
    python tools/stored_run.py [MiB ...]        e.g.  256 1024 4096
(XLZ_STORED_UNIT_KIB, in a library built with -DXLZ_DEV_KNOBS: the least size of a unit; 0: cuts at dictionary resets only)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch   # (before libxlz.so: torch brings its own HIP runtime, which must be the one that initialises the device)
torch.cuda.init()
import lzma_amd
from lzma_amd import Stream, FMT_LZMA2_RAW

CHUNK = 65536


def stored_stream(seed, n):
    """(raw LZMA2 stream: one 0x01 chunk, then 0x02 chunks of 64 KiB, end byte; its plaintext as a numpy array)"""
    assert n % CHUNK == 0
    p = np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)
    k = n // CHUNK
    s = np.empty(k * (CHUNK + 3) + 1, dtype=np.uint8)
    body = s[:-1].reshape(k, CHUNK + 3)
    body[:, 0] = 2
    body[0, 0] = 1
    body[:, 1] = 0xFF
    body[:, 2] = 0xFF
    body[:, 3:] = p.reshape(k, CHUNK)
    s[-1] = 0
    return s, p


def memcpy_rate(nbytes):
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * 5 / (e0.elapsed_time(e1) / 1e3)


sizes = [int(a) for a in sys.argv[1:]] or [512]
ctx = lzma_amd.Context(0)
for mib in sizes:
    n_streams = (mib + 3583) // 3584
    per = mib // n_streams
    made = [stored_stream(4242 + k, per << 20) for k in range(n_streams)]
    b = lzma_amd.Batch(ctx, [Stream(s.tobytes(), FMT_LZMA2_RAW, out_cap=p.size, dict_size=1 << 16) for s, p in made])
    b.run()
    b.sync()
    steps = 5
    ctx.event_record(0)
    for _ in range(steps):
        b.run()
    ctx.event_record(1)
    b.sync()
    ms = ctx.event_elapsed_ms(0, 1) / steps
    res = b.results()
    assert all(r[0] == p.size and r[1] == 0 for r, (_, p) in zip(res, made)), res
    for k, (_, p) in enumerate(made):
        assert hashlib.sha256(b.download(k, p.size)).digest() == hashlib.sha256(p.tobytes()).digest()
    cin, cout, units = b.stats()
    t0, t1, _ = b.unit_trace()
    slots, _ = b.launch_info()
    dur = (t1.astype(np.int64) - t0.astype(np.int64)) / 100.0
    span = float(t1.max() - t0.min()) / 100.0
    occ = float(dur.sum()) / (slots * span) if span else 0.0
    b.close()
    del b, made
    cp = memcpy_rate(per * n_streams << 20)
    print("XLZ_STORED_UNIT_KIB=%s: %d MiB in %d run(s) of stored chunks: %d units on %d slots (%.2f rounds), %.3f ms per launch, "
          "%.1f GiB/s decoded, %.3f of the 8 TB/s roof (algorithmic %d bytes = %.2f TB/s), slot occupancy %.2f, unit us p50 %.0f max %.0f, "
          "bit-exact; hipMemcpy device to device of the same bytes: %.2f TB/s read + written = %.3f of the roof"
          % (os.environ.get("XLZ_STORED_UNIT_KIB", "(default)"), per * n_streams, n_streams, units, slots, units / slots, ms,
             per * n_streams / 1024 / (ms / 1e3), (cin + cout) / (ms / 1e3) / 8e12, cin + cout, (cin + cout) / (ms / 1e3) / 1e12, occ,
             float(np.median(dur)), float(dur.max()), cp / 1e12, cp / 8e12), flush=True)
