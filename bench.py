#!/usr/bin/env python3
"""bench.py -- batched LZMA / LZMA2 decode throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (the gfx950 decode kernel) over one batch of synthetic
compressed streams that is already resident in HBM.

N = 1 (default): the headline `value` is BASELINE.json configs[1] ("cfg2-T": 4096 independent
LZMA1 streams, lc=3/lp=0/pb=2, 64 KiB dictionary, 1 MiB uncompressed each, text-like data),
timed over K steps.  The same JSON line carries a `configs` array with every other BASELINE
configuration measured in the same process on the same GPU (fewer steps each, see --side-steps):
cfg2-R (the incompressible family, the shape of the reference's own randomfile.dat benchmark),
cfg3 (65 536 streams), cfg4 (ONE raw LZMA2 stream of 4096 dictionary-reset units), cfg5
(8192 streams, lc2/lp1/pb1, 8 MiB dictionary) and cfg5-wrap (24 MiB streams whose 8 MiB window
wraps, distances up to 8 MiB).  Every decoded byte of every config is compared with the
plaintext's SHA-256, and every config has its own CPU baseline.

N > 1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of cfg3 -- ONE seeded
65 536-stream batch, rank r decodes shard r of `multigpu.partition_by_weight` (weights = the
uncompressed sizes), no data-path collective; torch.distributed only
provides the barrier and the max-over-ranks of the timed region.  value = bytes of the whole
batch / max-rank time.  The 1-GPU point of that curve is configs["cfg3"] of the N = 1 line.

Rank 0 prints ONE JSON line.  `roofline` prices the decode kernel against HBM bandwidth with
algorithmic bytes (compressed bytes read once + decoded bytes written once), timed with HIP
events on the kernel's own stream.  `cpu_baseline` times the CPU oracle (a C restatement of the
Go reference's algorithm; the Go toolchain does not exist here) on a bounded sample of the same
streams.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md chip table)

# liblzma encoder settings of the corpora.  The headline keeps round 1's preset 6; the side
# configs use a fast hash-chain setting so that ~25 GiB of plaintext is compressed within the
# run's time budget (ratio 0.35 instead of 0.31 on the text family: slightly more literals).
ENC_FAST = {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2}  # lzma.MODE_FAST, lzma.MF_HC3
ENC_FAST_NAME = "liblzma MODE_FAST/HC3/nice_len 32/depth 2"

CONFIGS = {
    "cfg2-T": dict(fmt="lzma1", family="T", streams=4096, size=1 << 20, dict=65536, lc=3, lp=0, pb=2, enc=6,
                   baseline="configs[1]: 4096 LZMA1 streams, 64 KiB dict, 1 MiB each"),
    "cfg2-R": dict(fmt="lzma1", family="R", streams=4096, size=1 << 20, dict=65536, lc=3, lp=0, pb=2, enc=ENC_FAST,
                   baseline="configs[1], incompressible family (literal-only, the shape of randomfile.dat.lzma)"),
    "cfg3": dict(fmt="lzma1", family="T", streams=65536, size=65536, dict=65536, lc=3, lp=0, pb=2, enc=ENC_FAST,
                 baseline="configs[2]: 65 536 LZMA1 streams, default params"),
    "cfg4": dict(fmt="lzma2", family="T", streams=1, segments=4096, size=256 << 10, dict=65536, lc=3, lp=0, pb=2,
                 enc=ENC_FAST, baseline="configs[3]: one large LZMA2 stream, 4096 dictionary-reset units"),
    "cfg5": dict(fmt="lzma1", family="T", streams=8192, size=2 << 20, dict=8 << 20, lc=2, lp=1, pb=1, enc=ENC_FAST,
                 baseline="configs[4]: 8192 LZMA1 streams, lc=2/lp=1/pb=1, 8 MiB dict"),
    "cfg5-wrap": dict(fmt="lzma1", family="F", streams=64, size=24 << 20, dict=8 << 20, lc=2, lp=1, pb=1, enc=ENC_FAST,
                      baseline="configs[4] variant: 24 MiB streams, the 8 MiB window wraps, distances up to 8 MiB"),
}
SIDE = ["cfg2-R", "cfg3", "cfg4", "cfg5", "cfg5-wrap"]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def effective_cpus():
    """Host CPUs this process may really use: the cgroup quota if there is one (a GPU box
    gives each GPU a share of the host's cores), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def enc_name(enc):
    return "liblzma preset %d" % enc if isinstance(enc, int) else ENC_FAST_NAME


def out_size_of(spec):
    return spec["size"] * (spec.get("segments", 1) if spec["fmt"] == "lzma2" else 1)


def workload_text(name, spec, per_gpu=True):
    n = spec["streams"]
    if spec["fmt"] == "lzma2":
        head = "%d raw LZMA2 stream(s)%s of %d dictionary-reset segments of %d B each" % (
            n, " per GPU" if per_gpu else "", spec["segments"], spec["size"])
    else:
        head = "%d independent LZMA1 (.lzma) streams%s" % (n, " per GPU" if per_gpu else "")
    return "%s: %s, lc=%d lp=%d pb=%d, %d KiB dict, %d B uncompressed per stream, family %s (corpus.py), %s, " \
           "inputs resident in HBM" % (name, head, spec["lc"], spec["lp"], spec["pb"], spec["dict"] >> 10,
                                       out_size_of(spec), spec["family"], enc_name(spec["enc"]))


# ---------------------------------------------------------------- corpus ----
def _gen_lzma1(job):
    import corpus
    family, seed, size, kw = job
    p = corpus.plain(family, seed, size)
    return corpus.compress_alone(p, **kw), hashlib.sha256(p).digest()


def _gen_lzma2_segment(job):
    import corpus
    family, seed, size, kw = job
    p = corpus.plain(family, seed, size)
    c = corpus.compress_raw_lzma2(p, **kw)
    assert c[-1] == 0
    return c[:-1], p


def make_corpus(pool, spec, base_seed, indices=None):
    """-> (compressed streams, sha256 digests of their plaintext) for the stream indices asked
    (default: all).  Stream i depends only on (base_seed, i): every rank of a multi-GPU run can
    make its own shard of the ONE seeded batch."""
    kw = dict(dict_size=spec["dict"], lc=spec["lc"], lp=spec["lp"], pb=spec["pb"], preset=spec["enc"])
    idx = list(range(spec["streams"])) if indices is None else list(indices)
    if spec["fmt"] == "lzma1":
        jobs = [(spec["family"], base_seed + i, spec["size"], kw) for i in idx]
        chunk = max(1, min(64, len(jobs) // 256))
        res = list(pool.map(_gen_lzma1, jobs, chunksize=chunk))
        return [r[0] for r in res], [r[1] for r in res]
    comp, dig = [], []
    for i in idx:  # few big streams: parallel over the segments of one stream
        jobs = [(spec["family"], (base_seed + i) * 4099 + k, spec["size"], kw) for k in range(spec["segments"])]
        parts = list(pool.map(_gen_lzma2_segment, jobs, chunksize=max(1, min(16, len(jobs) // 256))))
        h = hashlib.sha256()
        for _, pl in parts:
            h.update(pl)
        comp.append(b"".join(c for c, _ in parts) + b"\x00")
        dig.append(h.digest())
    return comp, dig


# ---------------------------------------------------------------- GPU leg ----
def make_batch(lzma_amd, ctx, spec, comp):
    osz = out_size_of(spec)
    if spec["fmt"] == "lzma2":
        return lzma_amd.Batch(ctx, [lzma_amd.Stream(c, lzma_amd.FMT_LZMA2_RAW, out_cap=osz, dict_size=spec["dict"])
                                    for c in comp])
    return lzma_amd.Batch(ctx, [lzma_amd.Stream(c, out_cap=osz) for c in comp])


def timed_steps(ctx, batch, steps, warmup, sync, barrier=None):
    """W untimed warm-up steps, then exactly K steps between barriers; -> (wall s, kernel ms/step)."""
    for _ in range(warmup):
        batch.run()
    batch.sync()
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    ctx.event_record(0)
    for _ in range(steps):
        batch.run()
    ctx.event_record(1)
    batch.sync()
    sync()
    t_local = time.perf_counter() - t0
    if barrier:
        barrier()
    return t_local, ctx.event_elapsed_ms(0, 1) / max(1, steps)  # HIP events on the kernel's stream


def verify_all(batch, n, osz, digests, tag):
    res = batch.results()
    bad = [i for i in range(n) if res[i][1] != 0 or res[i][0] != osz]
    if bad:
        raise SystemExit("%s: %d streams failed to decode (first: %d, status %d, out_len %d)"
                         % (tag, len(bad), bad[0], res[bad[0]][1], res[bad[0]][0]))
    t0 = time.time()

    def check(i):
        return hashlib.sha256(batch.download(i, osz)).digest() == digests[i]
    with ThreadPoolExecutor(max_workers=16) as ex:
        ok = list(ex.map(check, range(n)))
    if not all(ok):
        raise SystemExit("%s: decoded bytes differ from the plaintext for %d streams" % (tag, ok.count(False)))
    log("[%s] verified %d streams bit-exact (sha256 of every output byte) in %.1f s" % (tag, n, time.time() - t0))


def occupancy(batch, kernel_ms):
    """Slot occupancy of the persistent grid in the LAST launch, from the per-unit timestamps the
    waves take with s_memrealtime (100 MHz): busy wave-time / (slots x launch span).  The rest is
    tail: slots idling behind the slowest units of the last round."""
    import numpy as np
    t0, t1, in_len = batch.unit_trace()
    slots, lds = batch.launch_info()
    if len(t0) == 0 or slots == 0:
        return None
    dur = (t1.astype(np.int64) - t0.astype(np.int64)) / 100.0  # us
    span = float(t1.max() - t0.min()) / 100.0
    corr = float(np.corrcoef(in_len.astype(np.float64), dur)[0, 1]) if len(dur) > 2 and dur.std() > 0 else None
    q = np.percentile(dur, [0, 50, 90, 99, 100]) / 1e3
    return {"slot_occupancy": round(float(dur.sum()) / (slots * span), 4), "slots": int(slots), "lds_bytes_per_slot": int(lds),
            "units": int(len(dur)), "rounds": round(len(dur) / slots, 2), "launch_span_ms": round(span / 1e3, 3),
            "unit_ms": {"min": round(q[0], 3), "p50": round(q[1], 3), "p90": round(q[2], 3), "p99": round(q[3], 3),
                        "max": round(q[4], 3)},
            "corr_queue_key_vs_time": None if corr is None else round(corr, 3),
            "source": "live: per-unit s_memrealtime stamps of the last timed launch (xlz_batch_unit_trace); the "
                      "queue key is the unit's compressed size"}


def kernel_rev():
    """Identity of the kernel sources the numbers belong to (profiles are tied to it)."""
    h = hashlib.sha1()
    for f in ("xlz_kernel.hip", "xlz_fastpath.inc", "xlz_format.h"):
        h.update(open(os.path.join(ROOT, "lzma_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def profile_for(name):
    """Counters of the committed rocprofv3 --pmc passes over this config (tools/profile_bench.sh
    + tools/save_profile.py): bench.py cannot profile itself.  Reported only when the profile was
    taken on the kernel sources that are running now."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
    except Exception:
        return None
    e = pj.get(name)
    if not e or e.get("kernel_rev") != kernel_rev():
        return None
    return e


def roofline(name, cin, cout, units, kernel_ms, occ=None):
    algo = cin + cout  # per launch: compressed bytes read once + decoded bytes written once
    achieved = algo / 1e9 / (kernel_ms / 1e3)
    prof = profile_for(name)
    r = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 6),
         "traffic": prof["traffic_bytes_per_launch"] if prof else None,
         "kernel": "xlz::xlz_decode_kernel", "kernel_ms": round(kernel_ms, 3),
         "algorithmic_bytes_per_launch": algo, "units_per_launch": units}
    issue = dict(occ) if occ else {}
    if prof:
        r["traffic_from_profile"] = {"source": prof["source"], "kernel_rev": prof["kernel_rev"],
                                     "fetch": prof["fetch_bytes_per_launch_raw"], "write": prof["write_bytes_per_launch"]}
        if "issue" in prof:  # SQ instruction counters of the committed profile, per CU cycle
            issue["from_profile"] = prof["issue"]
    if issue:
        # the kernel is bound by instruction issue, three orders of magnitude under the HBM roof
        # (DESIGN.md section 3): this is the roofline a reader can act on
        r["issue"] = issue
    return r


# ---------------------------------------------------------------- CPU leg ----
def cpu_baseline(spec, comp, digests, threads, target_s):
    """The oracle on a bounded sample of the same streams, one stream per host thread."""
    import oracle
    osz = out_size_of(spec)
    if spec["fmt"] == "lzma2":
        # ONE stream: the reference's Reader2 is a single goroutine, so the honest baseline is one
        # core.  Sample = the first segments of the stream (cut at a dictionary-reset chunk).
        import corpus  # noqa: F401
        c = comp[0]
        seg_out = spec["size"]
        # walk chunk headers to the end of segment k
        want = max(1, min(spec["segments"], int(target_s * 80e6 / seg_out)))
        pos, out = 0, 0
        while out < want * seg_out:
            ctl = c[pos]
            if ctl >= 0x80:
                unc = (((ctl & 0x1F) << 16) | (c[pos + 1] << 8) | c[pos + 2]) + 1
                cs = ((c[pos + 3] << 8) | c[pos + 4]) + 1
                pos += (6 if ctl >= 0xC0 else 5) + cs
            else:
                unc = ((c[pos + 1] << 8) | c[pos + 2]) + 1
                pos += 3 + unc
            out += unc
        sample = [c[:pos] + b"\x00"]
        outs, sts, dt = oracle.decode_batch_mt(sample, [out], 1, fmt=2, dict_size=spec["dict"], timing=True)
        assert sts[0] == 0 and outs[0][1] == out
        return {"value": round(out / GIB / dt, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
                "sample": "the first %d of the stream's %d segments (%d MiB decoded) in %.2f s; ONE thread: the "
                          "reference's Reader2 is a single goroutine and cannot decode the units of one stream "
                          "concurrently; C restatement of the Go reference (oracle/xlz_oracle.c, gcc -O2); Go toolchain "
                          "absent" % (want, spec["segments"], out >> 20, dt)}
    n = len(comp)
    k = min(threads, n)
    _, _, per_round = oracle.decode_batch_mt(comp[:k], [osz] * k, threads, fmt=0, dict_size=spec["dict"], timing=True)
    rounds = max(1, int(target_s / max(per_round, 1e-3)))
    n_sample = min(n, max(k, rounds * k))
    outs, sts, dt = oracle.decode_batch_mt(comp[:n_sample], [osz] * n_sample, threads, fmt=0, dict_size=spec["dict"],
                                           timing=True)
    assert all(s == 0 for s in sts)
    for i in range(0, n_sample, max(1, n_sample // 16)):
        buf, n_out = outs[i]
        assert hashlib.sha256(buf.raw[:n_out]).digest() == digests[i]
    per_core = n_sample * osz / (1 << 20) / dt / min(threads, n_sample)
    import shutil
    return {"value": round(n_sample * osz / GIB / dt, 4), "unit": "GiB/s", "cores": min(threads, n_sample), "kind": "port",
            "per_core_mib_s": round(per_core, 1),
            "go_toolchain_on_this_box": bool(shutil.which("go") or shutil.which("gccgo")),  # probed every run
            "sample": "%d of the %d streams (%d MiB decoded) in %.2f s of decode wall time; C restatement of the Go "
                      "reference's algorithm (oracle/xlz_oracle.c, gcc -O2), one stream per thread on %d host CPUs "
                      "(%.0f MiB/s per core; the reference publishes 42.59 MiB/s on tar data and 16.47 MB/s on random "
                      "data for one 2.6 GHz i7 core, ReadMe.md:10-20, reader1_test.go:109-114); Go toolchain absent"
                      % (n_sample, n, n_sample * osz >> 20, dt, min(threads, n_sample), per_core)}


def liblzma_sanity(comp, osz, threads, target_s=3.0):
    """liblzma (Python's lzma module, GIL released while decoding) on the
    same streams, one per thread -- a second, independent CPU decoder for scale."""
    import lzma
    n = max(threads, min(len(comp), int(target_s * threads * 60e6 / osz)))
    n = min(n, len(comp))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        sizes = list(ex.map(lambda c: len(lzma.decompress(c, format=lzma.FORMAT_ALONE)), comp[:n]))
    dt = time.perf_counter() - t0
    assert all(s == osz for s in sizes)
    return {"value": round(n * osz / GIB / dt, 4), "unit": "GiB/s", "cores": threads, "kind": "liblzma 5.x (xz) via Python",
            "sample": "%d streams in %.2f s" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--side-steps", type=int, default=3, help="timed steps of each entry of `configs` (1 warm-up)")
    ap.add_argument("--configs", default="all",
                    help="N=1: comma list of side configs to run (%s), 'all' or 'none'" % ",".join(SIDE))
    ap.add_argument("--headline", default="cfg2-T", choices=list(CONFIGS), help="N=1: the config reported as `value`")
    ap.add_argument("--scale", type=float, default=1.0, help="dev runs: multiply every config's stream count")
    ap.add_argument("--trace-out", default="", help="dev: save the per-unit (t_start, t_end, in_len) stamps of every config "
                                                     "as <prefix><config>.npz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-target-s", type=float, default=12.0)
    ap.add_argument("--side-cpu-target-s", type=float, default=4.0)
    args = ap.parse_args()

    from lzma_amd import multigpu
    rank, world, local_rank = multigpu.env_rank()
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))

    specs = {k: dict(v) for k, v in CONFIGS.items()}
    if args.scale != 1.0:
        for v in specs.values():
            if v["fmt"] == "lzma2":
                v["segments"] = max(8, int(v["segments"] * args.scale))
            else:
                v["streams"] = max(8, int(v["streams"] * args.scale))
    if world == 1:
        side = [] if args.configs == "none" else SIDE if args.configs == "all" else \
            [c for c in args.configs.split(",") if c and c != args.headline]
        names = [args.headline] + [c for c in side if c != args.headline]
    else:
        names = ["cfg3"]

    # ---- synthetic corpora, generated BEFORE anything touches the GPU (the generator forks
    # worker processes; a process that has initialised HIP must not fork workers)
    ncpu = effective_cpus()
    workers = max(1, min((os.cpu_count() or 1) // max(1, min(world, 8)), 64))
    corp = {}
    shard = None
    t_gen0 = time.time()
    with ProcessPoolExecutor(max_workers=workers) as pool:
        for name in names:
            spec = specs[name]
            t0 = time.time()
            if world > 1:
                # strong scaling: ONE batch (seed 1), this rank's shard of it
                weights = [out_size_of(spec)] * spec["streams"]
                shard = multigpu.partition_by_weight(weights, world)[rank]
                comp, dig = make_corpus(pool, spec, 1, shard)
            else:
                comp, dig = make_corpus(pool, spec, 1)
            corp[name] = (comp, dig)
            log("[rank %d] corpus %s: %d streams x %d B, ratio %.3f, generated in %.1f s with %d workers"
                % (rank, name, len(comp), out_size_of(spec), sum(map(len, comp)) / (len(comp) * out_size_of(spec)),
                   time.time() - t0, workers))
    gen_s = time.time() - t_gen0

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decode path has no CPU fallback)")
    # rehearsal knobs (one-GPU box): XLZ_BENCH_DEVICE pins every rank to one device,
    # XLZ_BENCH_BACKEND=gloo replaces RCCL for the barrier / max-reduce (RCCL refuses two ranks per GPU)
    if os.environ.get("XLZ_BENCH_DEVICE"):
        local_rank = int(os.environ["XLZ_BENCH_DEVICE"])
    backend = os.environ.get("XLZ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import lzma_amd
    from lzma_amd import build
    build.build()
    ctx = lzma_amd.Context(local_rank)

    def barrier():
        multigpu.barrier(dist, torch.cuda.synchronize)

    def gpu_leg(name, steps, warmup):
        spec = specs[name]
        comp, dig = corp[name]
        osz = out_size_of(spec)
        t0 = time.time()
        batch = make_batch(lzma_amd, ctx, spec, comp)
        log("[rank %d] %s: batch created + uploaded in %.1f s" % (rank, name, time.time() - t0))
        t_local, kernel_ms = timed_steps(ctx, batch, steps, warmup, torch.cuda.synchronize, barrier if world > 1 else None)
        verify_all(batch, len(comp), osz, dig, "rank %d %s" % (rank, name))
        cin, cout, units = batch.stats()
        occ = occupancy(batch, kernel_ms)
        if args.trace_out:
            import numpy as np
            t0_, t1_, il_ = batch.unit_trace()
            np.savez_compressed(args.trace_out + name + ".npz", t_start=t0_, t_end=t1_, in_len=il_)
        batch.close()
        return t_local, kernel_ms, cin, cout, units, occ

    def cpu_leg(name, target_s):
        if args.no_cpu_baseline:
            return None
        spec = specs[name]
        comp, dig = corp[name]
        threads = int(os.environ.get("XLZ_BENCH_CPU_THREADS", "0")) or ncpu
        return cpu_baseline(spec, comp, dig, threads, target_s)

    # ------------------------------------------------------------ N > 1: strong scaling ----
    if world > 1:
        name = "cfg3"
        spec = specs[name]
        t_local, kernel_ms, cin, cout, units, occ = gpu_leg(name, args.steps, args.warmup)
        t_max = multigpu.max_over_ranks(t_local, dist, device="cuda" if backend == "nccl" else "cpu")
        n_mine = multigpu.max_over_ranks(float(len(corp[name][0])), dist, device="cuda" if backend == "nccl" else "cpu")
        sums = torch.tensor([float(len(corp[name][0])), float(cin), float(cout)], dtype=torch.float64,
                            device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(sums)  # 24 bytes of bookkeeping, not data: streams / bytes over all shards
        cpu = cpu_leg(name, args.side_cpu_target_s) if rank == 0 else None
        barrier()
        if rank == 0:
            total_streams = int(sums[0].item())
            assert total_streams == spec["streams"], "the shards do not add up to the batch"
            total_out = spec["streams"] * out_size_of(spec) * args.steps
            line = {
                "metric": "decompressed GiB/s (aggregate batch)", "value": round(total_out / GIB / t_max, 4), "unit": "GiB/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(t_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": workload_text(name, spec, per_gpu=False) + "; ONE batch split over %d GPUs" % world,
                           "streams_total": total_streams, "streams_largest_shard": int(n_mine),
                           "bytes_per_stream": out_size_of(spec),
                           "compression_ratio": round(sums[1].item() / sums[2].item(), 4), "bit_exact": "all",
                           "parallelism": "one batch sharded by stream x%d (partition_by_weight), no collective" % world,
                           "n1_reference": "configs[cfg3] of the --gpus 1 line is the 1-GPU point of this curve"},
                "roofline": dict(roofline(name, cin, cout, units, kernel_ms, occ), note="rank 0's shard"),
                "cpu_baseline": cpu,
            }
            print(json.dumps(line), flush=True)
        dist.destroy_process_group()
        return

    # ------------------------------------------------------------ N = 1: all configs ----
    results = {}
    for name in names:
        head = name == args.headline
        steps, warmup = (args.steps, args.warmup) if head else (args.side_steps, 1)
        t_local, kernel_ms, cin, cout, units, occ = gpu_leg(name, steps, warmup)
        spec = specs[name]
        n = len(corp[name][0])
        total_out = n * out_size_of(spec) * steps
        results[name] = {
            "name": name, "baseline_config": spec["baseline"], "workload": workload_text(name, spec),
            "value": round(total_out / GIB / t_local, 4), "unit": "GiB/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(t_local / steps * 1e3, 3), "kernel_ms": round(kernel_ms, 3),
            "streams": n, "bytes_per_stream": out_size_of(spec), "compression_ratio": round(cin / cout, 4),
            "bit_exact": "all", "roofline": roofline(name, cin, cout, units, kernel_ms, occ),
        }
    for name in names:  # CPU legs after all GPU work: the host cores are quiet
        head = name == args.headline
        results[name]["cpu_baseline"] = cpu_leg(name, args.cpu_target_s if head else args.side_cpu_target_s)
        log("[cpu] %s: %s" % (name, results[name]["cpu_baseline"] and results[name]["cpu_baseline"]["value"]))
    h = results[args.headline]
    spec = specs[args.headline]
    sanity = None
    if not args.no_cpu_baseline and spec["fmt"] == "lzma1":
        try:
            sanity = liblzma_sanity(corp[args.headline][0], out_size_of(spec), ncpu)
        except Exception as e:  # a sanity line must never fail the bench
            log("liblzma sanity line skipped: %r" % (e,))
    line = {
        "metric": "decompressed GiB/s (aggregate batch)", "value": h["value"], "unit": "GiB/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": h["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": h["workload"], "streams_per_gpu": h["streams"], "bytes_per_stream": h["bytes_per_stream"],
                   "compression_ratio": h["compression_ratio"], "bit_exact": "all",
                   "parallelism": "shard-by-stream x1, no collective", "corpus_generation_s": round(gen_s, 1),
                   "kernel_rev": kernel_rev()},
        "roofline": h["roofline"],
        "cpu_baseline": h["cpu_baseline"],
        "cpu_sanity_liblzma": sanity,
        "configs": [results[n] for n in names if n != args.headline],
    }
    print(json.dumps(line), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
