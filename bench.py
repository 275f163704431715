#!/usr/bin/env python3
"""bench.py -- batched LZMA / LZMA2 decode throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (the gfx950 decode kernel) over one batch of synthetic
compressed streams that is already resident in HBM.

ONE workload at every N (strong scaling): BASELINE.json configs[2] / north_star's target, "cfg3" -- ONE
seeded batch of 65 536 independent LZMA1 streams (lc=3/lp=0/pb=2, 64 KiB dictionary, 64 KiB of
text-like data each).  Rank r of N (one rank per GPU under torch.distributed.run) generates and
decodes shard r of `multigpu.partition_by_weight` (weights = the uncompressed sizes, SURVEY.md section
8e); N = 1 decodes the whole batch.  No data-path collective: torch.distributed provides the barrier,
the MAX of the timed region and a 24-byte SUM of bookkeeping.  value = bytes of the whole batch x K /
max-rank time.  (`--headline cfg3-heavy`: the same batch with 1 MiB per stream, SURVEY 8d's optional
heavy variant for an 8-GPU node -- 8 x 8192 streams of 64 KiB are only two wave rounds per GPU.)

The LAST line of stdout is ONE compact JSON object (< 4 KB: tests/test_bench_host.py) -- metric / value / config /
roofline / cpu_baseline for the headline plus one row per side config.  Everything else goes to a sidecar file,
`bench_detail.json` (--detail-out; also copied under gpurun_out/ when that directory exists), and a summary to stderr:
  * `configs`: every other BASELINE configuration measured in the same process on the same GPU
    (--side-steps each): cfg2-T (configs[1]: 4096 x 1 MiB), cfg2-R (incompressible, the shape of the reference's
    randomfile.dat benchmark), cfg4 (ONE raw LZMA2 stream of 4096 dictionary-reset units), cfg4-R (the same with
    incompressible segments: stored chunks only, the shape of randomfile.dat.lzma2), cfg5 (8192
    streams, lc2/lp1/pb1, 8 MiB dictionary), cfg5-wrap (24 MiB streams whose 8 MiB window wraps); on request
    (--configs) cfg2-T-p6, the same plaintext as cfg2-T behind liblzma preset 6 (rounds 1 and 2's headline).
    All corpora use ONE encoder setting (ENC_FAST) so that the configs can be compared; every config
    has its own `roofline` (incl. `issue`: the instruction-issue and lone-wave-latency bounds the LZMA
    paths really run against, and the HBM roof in decoded bytes, which binds the stored-chunk config) and
    `cpu_baseline`; every decoded byte is compared with the plaintext's SHA-256;
  * `host_to_host`: cfg3 and cfg2-T through xlz_decode_batch -- host buffers in, host buffers out,
    PCIe included, with the phase times (what a Go caller of the drop-in sees; never `value`);
  * `stream_count_sweep`: 64 / 256 / 1024 / 4096 streams of cfg2-T with the CPU baseline beside each:
    one wave decodes one stream, so a small batch leaves the chip idle -- the break-even is stated;
  * `containers`: a multi-block .xz file through xlz_xz_decode (host to host, CRC64 verified).

The corpora are compressed by a pool of worker processes forked BEFORE anything touches the GPU.  Only the headline's
corpus is waited for up front; the side corpora are compressed while the GPU legs run, but never inside the headline's
timed region or a host-timed (PCIe-inclusive) leg: the pool is drained before those, and side configs report the
HIP-event time of their launches.

`roofline` prices the decode kernel against HBM bandwidth with algorithmic bytes (compressed bytes
read once + decoded bytes written once), timed with HIP events on the kernel's own stream.
`cpu_baseline` times the CPU oracle (a C restatement of the Go reference's algorithm; the Go toolchain
does not exist here) on a bounded sample of the same streams, on rank 0 at N = 1.
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

# liblzma encoder setting of the corpora: ONE fast hash-chain setting for every config, so that ~35 GiB of plaintext is
# compressed within the run's time budget and the configs can be compared with each other (round 2 mixed preset 6 for
# the headline with this one for the rest).  cfg2-T-p6 keeps rounds 1-2's headline (the same plaintext behind preset 6:
# ratio 0.31 instead of 0.35, fewer literals per byte) for continuity.
ENC_FAST = {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2}  # lzma.MODE_FAST, lzma.MF_HC3
ENC_FAST_NAME = "liblzma MODE_FAST/HC3/nice_len 32/depth 2"

CONFIGS = {
    "cfg3": dict(fmt="lzma1", family="T", streams=65536, size=65536, dict=65536, lc=3, lp=0, pb=2, enc=ENC_FAST,
                 baseline="configs[2]: 65 536 LZMA1 streams, default params (north_star's target batch)"),
    "cfg3-heavy": dict(fmt="lzma1", family="T", streams=65536, size=1 << 20, dict=65536, lc=3, lp=0, pb=2, enc=ENC_FAST,
                       baseline="configs[2], heavy variant (SURVEY 8d): 65 536 streams of 1 MiB, for 8-GPU nodes"),
    "cfg2-T": dict(fmt="lzma1", family="T", streams=4096, size=1 << 20, dict=65536, lc=3, lp=0, pb=2, enc=ENC_FAST,
                   baseline="configs[1]: 4096 LZMA1 streams, 64 KiB dict, 1 MiB each"),
    "cfg2-T-p6": dict(fmt="lzma1", family="T", streams=4096, size=1 << 20, dict=65536, lc=3, lp=0, pb=2, enc=6,
                      baseline="configs[1] behind liblzma preset 6 (the headline of rounds 1 and 2)"),
    "cfg2-R": dict(fmt="lzma1", family="R", streams=4096, size=1 << 20, dict=65536, lc=3, lp=0, pb=2, enc=ENC_FAST,
                   baseline="configs[1], incompressible family (literal-only, the shape of randomfile.dat.lzma)"),
    "cfg4": dict(fmt="lzma2", family="T", streams=1, segments=4096, size=256 << 10, dict=65536, lc=3, lp=0, pb=2,
                 enc=ENC_FAST, baseline="configs[3]: one large LZMA2 stream, 4096 dictionary-reset units"),
    "cfg4-R": dict(fmt="lzma2", family="R", streams=1, segments=4096, size=256 << 10, dict=65536, lc=3, lp=0, pb=2,
                   enc=ENC_FAST, baseline="configs[3], incompressible variant: stored chunks only, the shape of the reference's "
                                          "own LZMA2 benchmark file randomfile.dat.lzma2 (reader2_test.go:31-36) -- the one sub-path "
                                          "that is a plain copy"),
    "cfg5": dict(fmt="lzma1", family="T", streams=8192, size=2 << 20, dict=8 << 20, lc=2, lp=1, pb=1, enc=ENC_FAST,
                 baseline="configs[4]: 8192 LZMA1 streams, lc=2/lp=1/pb=1, 8 MiB dict"),
    "cfg5-wrap": dict(fmt="lzma1", family="F", streams=64, size=24 << 20, dict=8 << 20, lc=2, lp=1, pb=1, enc=ENC_FAST,
                      baseline="configs[4] variant: 24 MiB streams, the 8 MiB window wraps, distances up to 8 MiB"),
}
HEADLINES = ["cfg3", "cfg3-heavy"]
SIDE = ["cfg2-T", "cfg2-R", "cfg4", "cfg4-R", "cfg5", "cfg5-wrap"]   # default side list (cfg2-T-p6: on request, its preset-6 corpus takes 40 s)
SIDE_ALL = SIDE + ["cfg2-T-p6"]
EXTRAS = ["h2h", "sweep", "xz", "lone", "shard"]   # lone: the 64-unit launches behind roofline.issue.latency_bound;
# shard: rank 0's shard of the headline batch at N = 2, 4, 8 decoded on THIS GPU -- a projection of the strong-scaling curve
SHARD_COUNTS = [2, 4, 8]
SWEEP_COUNTS = [64, 256, 1024]  # (4096 is cfg2-T itself)


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


class PendingCorpus:
    """A corpus whose compression jobs have been handed to the pool (ProcessPoolExecutor.map submits every job at
    once and yields the results in order); result() waits for them."""

    def __init__(self, spec, parts):
        self.spec, self.parts, self.done = spec, parts, None
        self.t0 = time.time()

    def result(self):
        if self.done is None:
            if self.spec["fmt"] == "lzma1":
                res = list(self.parts)
                self.done = [r[0] for r in res], [r[1] for r in res]
            else:
                comp, dig = [], []
                for it in self.parts:  # few big streams: the jobs are the segments of one stream
                    parts = list(it)
                    h = hashlib.sha256()
                    for _, pl in parts:
                        h.update(pl)
                    comp.append(b"".join(c for c, _ in parts) + b"\x00")
                    dig.append(h.digest())
                self.done = comp, dig
            self.parts = None
            self.seconds = time.time() - self.t0
        return self.done


def start_corpus(pool, spec, base_seed, indices=None):
    """Hands the compression jobs of a corpus to the pool and returns at once (PendingCorpus).  Stream i depends
    only on (base_seed, i): every rank of a multi-GPU run can make its own shard of the ONE seeded batch."""
    kw = dict(dict_size=spec["dict"], lc=spec["lc"], lp=spec["lp"], pb=spec["pb"], preset=spec["enc"])
    idx = list(range(spec["streams"])) if indices is None else list(indices)
    if spec["fmt"] == "lzma1":
        jobs = [(spec["family"], base_seed + i, spec["size"], kw) for i in idx]
        chunk = max(1, min(64, len(jobs) // 256))
        return PendingCorpus(spec, pool.map(_gen_lzma1, jobs, chunksize=chunk))
    its = []
    for i in idx:
        jobs = [(spec["family"], (base_seed + i) * 4099 + k, spec["size"], kw) for k in range(spec["segments"])]
        its.append(pool.map(_gen_lzma2_segment, jobs, chunksize=max(1, min(16, len(jobs) // 256))))
    return PendingCorpus(spec, its)


def make_corpus(pool, spec, base_seed, indices=None):
    """-> (compressed streams, sha256 digests of their plaintext) for the stream indices asked (default: all)."""
    return start_corpus(pool, spec, base_seed, indices).result()


# ---------------------------------------------------------------- GPU leg ----
def make_batch(lzma_amd, ctx, spec, comp):
    osz = out_size_of(spec)
    if spec["fmt"] == "lzma2":
        return lzma_amd.Batch(ctx, [lzma_amd.Stream(c, lzma_amd.FMT_LZMA2_RAW, out_cap=osz, dict_size=spec["dict"])
                                    for c in comp])
    return lzma_amd.Batch(ctx, [lzma_amd.Stream(c, out_cap=osz) for c in comp])


EVENT_SLOTS = 64  # xlz_ctx_event_record slots (include/xlz.h)


def timed_steps(ctx, batch, steps, warmup, sync, barrier=None):
    """W untimed warm-up steps, then exactly K steps between barriers; -> (wall s, kernel ms/step, per-step ms list).
    HIP events on the kernel's own stream: one in front of the first step and one behind every step (as many as
    the context has slots for), so that the MEDIAN step is reported beside the mean."""
    for _ in range(warmup):
        batch.run()
    batch.sync()
    if barrier:
        barrier()
    sync()
    per_step = steps <= EVENT_SLOTS - 1
    t0 = time.perf_counter()
    ctx.event_record(0)
    for k in range(steps):
        batch.run()
        if per_step and k + 1 < steps:
            ctx.event_record(k + 1)
    ctx.event_record(steps if per_step else 1)
    batch.sync()
    sync()
    t_local = time.perf_counter() - t0
    if barrier:
        barrier()
    last = steps if per_step else 1
    each = [ctx.event_elapsed_ms(k, k + 1) for k in range(steps)] if per_step else []
    return t_local, ctx.event_elapsed_ms(0, last) / max(1, steps), each


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return None if n == 0 else xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


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
            "kernel": batch.kernel_name(),
            "units": int(len(dur)), "rounds": round(len(dur) / slots, 2), "launch_span_ms": round(span / 1e3, 3),
            "unit_ms": {"min": round(q[0], 3), "p50": round(q[1], 3), "p90": round(q[2], 3), "p99": round(q[3], 3),
                        "max": round(q[4], 3)},
            "corr_queue_key_vs_time": None if corr is None else round(corr, 3),
            "source": "live: per-unit s_memrealtime stamps of the last timed launch (xlz_batch_unit_trace); the "
                      "queue key is the unit's compressed size"}


def library():
    """which binary runs: path, SHA-256 of the file, the source hash compiled into it (xlz_build_id) and whether that is
    the hash of the sources in the tree"""
    from lzma_amd import _native
    return _native.library_info()


def kernel_rev():
    """Identity of the kernel the numbers belong to: the hash of the device code's sources compiled into the loaded
    library (profiles are tied to it; host-side changes do not change it)."""
    return library()["kernel_id"]


def profile_for(name):
    """Counters of the committed rocprofv3 --pmc passes over this config (tools/profile_bench.sh
    + tools/save_profile.py): bench.py cannot profile itself.  Reported only when the profile was
    taken on the library build that is running now."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
    except Exception:
        return None
    e = pj.get(name)
    if not e or e.get("kernel_rev") != kernel_rev():
        return None
    return e


# Measured ceilings of one CU with 16 single-wave workgroups (tools/ubench/mix2.hip, profiles/r02/ubench_mix2.txt):
# scalar instructions share ONE port per CU (0.97 per cycle), simple wave64 VALU instructions reach 1.28 per cycle.
SALU_PER_CU_CYCLE, VALU_PER_CU_CYCLE, CUS, CLOCK_HZ = 0.97, 1.28, 256, 2.4e9


def issue_bounds(prof, out_bytes_per_launch, kernel_ms, lone, slots, algo_bytes_per_launch=None):
    """The two bounds this kernel really runs against (the HBM roof is three orders of magnitude away):
      issue_bound    the CU's instruction ports: per decoded byte the kernel issues s scalar and v vector instructions
                     (SQ counters of the committed profile of THIS build and config), a CU issues at most 0.97 / 1.28 of
                     them per cycle -> CUs x clock / max(s / 0.97, v / 1.28) bytes per second;
      latency_bound  one wave's serial dependent chain: the rate of a LONE wave on this data (a live 64-unit launch, one
                     wave on 64 of 1024 SIMDs: nothing competes for issue) x the wave slots of the full launch.
      hbm_bound      the contract's roof expressed in decoded bytes: 8 TB/s x decoded / algorithmic bytes.  Three orders of
                     magnitude away for the LZMA paths; THE bound of a launch of stored LZMA2 chunks (cfg4-R: a plain copy).
    frac_of_bound = achieved / min(all).  The LZMA kernel sits at the knee of the first two with 16 waves per CU."""
    achieved = out_bytes_per_launch / (kernel_ms / 1e3)
    r = {"achieved_decoded_GBps": round(achieved / 1e9, 3)}
    bounds = []
    if prof and "issue" in prof and "salu_per_decoded_byte" in prof["issue"]:
        i = prof["issue"]
        cyc = max(i["salu_per_decoded_byte"] / SALU_PER_CU_CYCLE, i["valu_per_decoded_byte"] / VALU_PER_CU_CYCLE)
        ib = CUS * CLOCK_HZ / cyc
        r["issue_bound_GBps"] = round(ib / 1e9, 3)
        r["issue_bound_from"] = {"salu_per_decoded_byte": i["salu_per_decoded_byte"], "valu_per_decoded_byte": i["valu_per_decoded_byte"],
                                 "branch_per_decoded_byte": i.get("branch_per_decoded_byte"),
                                 "ceilings_per_cu_cycle": {"salu": SALU_PER_CU_CYCLE, "valu": VALU_PER_CU_CYCLE},
                                 "cus": CUS, "clock_ghz": CLOCK_HZ / 1e9, "binding_port": "salu" if
                                 i["salu_per_decoded_byte"] / SALU_PER_CU_CYCLE >= i["valu_per_decoded_byte"] / VALU_PER_CU_CYCLE else "valu",
                                 "source": i.get("source")}
        bounds.append(ib)
    if lone and slots:
        lb = lone["bytes_per_s_per_wave"] * slots
        r["latency_bound_GBps"] = round(lb / 1e9, 3)
        r["latency_bound_from"] = dict(lone, wave_slots=slots)
        bounds.append(lb)
    names = (["issue"] if "issue_bound_GBps" in r else []) + (["latency"] if "latency_bound_GBps" in r else [])
    if algo_bytes_per_launch:
        hb = HBM_PEAK_GBS * 1e9 * out_bytes_per_launch / algo_bytes_per_launch
        r["hbm_bound_GBps"] = round(hb / 1e9, 3)
        bounds.append(hb)
        names.append("hbm")
    if bounds:
        k = min(range(len(bounds)), key=lambda j: bounds[j])
        r["frac_of_bound"] = round(achieved / bounds[k], 4)
        r["binding"] = names[k]
        r["bounds_known"] = names
    return r


def roofline(name, cin, cout, units, kernel_ms, occ=None, lone=None):
    algo = cin + cout  # per launch: compressed bytes read once + decoded bytes written once
    achieved = algo / 1e9 / (kernel_ms / 1e3)
    prof = profile_for(name)
    r = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 6),
         "traffic": prof["traffic_bytes_per_launch"] if prof else None,
         "kernel": (occ or {}).get("kernel", "xlz::xlz_decode_kernel"), "kernel_ms": round(kernel_ms, 3),
         "algorithmic_bytes_per_launch": algo, "units_per_launch": units}
    issue = dict(occ) if occ else {}
    issue.update(issue_bounds(prof, cout, kernel_ms, lone, occ["slots"] if occ else None, algo))
    if prof:
        r["traffic_from_profile"] = {"source": prof["source"], "kernel_rev": prof["kernel_rev"],
                                     "fetch": prof["fetch_bytes_per_launch_raw"], "write": prof["write_bytes_per_launch"],
                                     "fetch_correction": prof.get("fetch_correction", 1)}
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
                "sample_short": "first %d of %d segments (%d MiB) in %.1f s on ONE thread (a Reader2 is one goroutine); C restatement of "
                                "the Go reference (oracle/xlz_oracle.c); Go toolchain absent" % (want, spec["segments"], out >> 20, dt),
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
            "sample_short": "%d of %d streams (%d MiB) in %.1f s, one stream per thread on %d host threads; C restatement of the Go "
                            "reference (oracle/xlz_oracle.c, gcc -O2); Go toolchain absent" % (n_sample, n, n_sample * osz >> 20, dt, min(threads, n_sample)),
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


def lone_wave_sample(spec, comp):
    """the first 64 units of a corpus as a batch of their own: 64 waves on 1024 SIMDs, nothing competes for issue"""
    if spec["fmt"] == "lzma2":   # the first 64 segments of the stream (cut at a dictionary-reset chunk)
        c, pos, out, want = comp[0], 0, 0, min(64, spec["segments"]) * spec["size"]
        while out < want:
            ctl = c[pos]
            if ctl >= 0x80:
                unc = (((ctl & 0x1F) << 16) | (c[pos + 1] << 8) | c[pos + 2]) + 1
                pos += (6 if ctl >= 0xC0 else 5) + ((c[pos + 3] << 8) | c[pos + 4]) + 1
            else:
                unc = ((c[pos + 1] << 8) | c[pos + 2]) + 1
                pos += 3 + unc
            out += unc
        return [c[:pos] + b"\x00"], out
    return comp[:64], spec["size"]


def xz_file(pool, blocks, size):
    """-> (bytes of a multi-block .xz file, sha256 of the plaintext): `blocks` independently compressed blocks as
    concatenated single-block streams (what pixz / `xz -T` style tools write; CRC64 checks)"""
    res = list(pool.map(_gen_xz_block, [(9000 + i, size) for i in range(blocks)], chunksize=max(1, blocks // 256)))
    h = hashlib.sha256()
    for _, pl in res:
        h.update(pl)
    return b"".join(c for c, _ in res), h.digest()


def _gen_xz_block(job):
    import lzma
    import corpus
    seed, size = job
    p = corpus.plain("T", seed, size)
    filt = [dict(corpus.lzma1_filters(dict_size=1 << 20, preset=ENC_FAST)[0], id=lzma.FILTER_LZMA2)]
    return lzma.compress(p, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64, filters=filt), p


def _warm(_):
    time.sleep(0.05)
    return os.getpid()


def short_cpu(cpu):
    """the cpu_baseline object of the compact line: the contract's fields, the sample in one sentence"""
    if not cpu:
        return None
    r = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "per_core_mib_s") if k in cpu}
    r["sample"] = cpu.get("sample_short") or cpu.get("sample", "")[:200]
    return r


def compact_line(full, detail_path=None):
    """The ONE line the driver parses (the round-3 line had grown to 32 KB and was not parsed): the
    contract's keys, the headline's config / roofline / cpu_baseline, one row per side config.  `full` is the detail
    record (what rounds 1-3 printed); everything not copied here stays in the sidecar file."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in keys}
    cfg = full["config"]
    line["config"] = {k: cfg.get(k) for k in ("workload", "streams_total", "streams_largest_shard", "bytes_per_stream",
                                              "compression_ratio", "bit_exact", "parallelism", "kernel_rev")}

    def rf_short(rf):
        r = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "kernel_ms_median",
                                    "algorithmic_bytes_per_launch", "units_per_launch")}
        iss = rf.get("issue") or {}
        for k in ("frac_of_bound", "binding", "issue_bound_GBps", "latency_bound_GBps", "hbm_bound_GBps", "slot_occupancy"):
            if k in iss:
                r[k] = iss[k]
        fp = iss.get("from_profile") or {}
        if "instructions_per_decoded_byte" in fp:
            r["instructions_per_decoded_byte"] = fp["instructions_per_decoded_byte"]
        if rf.get("note"):
            r["note"] = rf["note"]
        return r
    line["roofline"] = rf_short(full["roofline"])
    line["cpu_baseline"] = short_cpu(full.get("cpu_baseline"))
    rows = []
    for c in full.get("configs") or []:
        iss = c["roofline"].get("issue") or {}
        cpu = c.get("cpu_baseline") or {}
        rows.append({"name": c["name"], "value": c["value"], "value_wall": c.get("value_wall"), "kernel_ms": c["kernel_ms"], "frac": c["roofline"]["frac"],
                     "frac_of_bound": iss.get("frac_of_bound"), "binding": iss.get("binding"),
                     "cpu": cpu.get("value"), "cpu_cores": cpu.get("cores")})
    if rows:
        line["configs"] = rows
        line["configs_unit"] = ("GiB/s decoded, device-resident, every byte verified; value = bytes / HIP-event time of the timed launches, "
                                "value_wall = bytes / host wall time of the same steps (the headline's `value` is wall time); "
                                "cpu = the oracle on cpu_cores host threads")
    if full.get("host_to_host"):
        line["host_to_host"] = {x["name"]: x["value"] for x in full["host_to_host"]}
    sw = full.get("stream_count_sweep")
    if isinstance(sw, dict):
        line["break_even_streams"] = sw.get("break_even_streams")
    if full.get("containers"):
        line["xz_1024_blocks_host_to_host"] = full["containers"][0]["value"]
    if full.get("scaling_projection"):
        line["scaling_projection"] = {"label": "one GPU, 1/N shard of the same batch (no inter-GPU traffic in this path)",
                                      "points": [[x["n_gpus"], x["value_projected"], x["efficiency_projected"]] for x in full["scaling_projection"]["points"]],
                                      "columns": ["n_gpus", "GiB/s projected", "of linear"]}
    if full.get("timing"):
        line["bench_wall_s"] = full["timing"].get("total_s")
    if detail_path:
        line["detail"] = detail_path
    return line


def write_detail(full, path):
    """the full record (what rounds 1-3 printed as one 32 KB line) next to the bench; a copy under gpurun_out/ when
    that directory exists, so that a gpurun call brings it home"""
    written = []
    go = os.path.join(ROOT, "gpurun_out")
    inside = os.path.abspath(path).startswith(os.path.abspath(go) + os.sep)
    for p in [path] + ([os.path.join(go, os.path.basename(path))] if os.path.isdir(go) and not inside else []):
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
            written.append(p)
        except OSError as e:
            log("bench_detail: cannot write %s: %r" % (p, e))
    return written


def main():
    t_main0 = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--side-steps", type=int, default=3, help="timed steps of each entry of `configs` (1 warm-up)")
    ap.add_argument("--configs", default="all",
                    help="N=1: comma list of side configs to run (%s), 'all' (= %s) or 'none'" % (",".join(SIDE_ALL), ",".join(SIDE)))
    ap.add_argument("--extras", default="all", help="N=1: comma list of %s, 'all' or 'none'" % ",".join(EXTRAS))
    ap.add_argument("--headline", default="cfg3", choices=list(CONFIGS),
                    help="the ONE workload reported as `value` at every N (cfg3; cfg3-heavy for big nodes; any other "
                         "config for profiling runs)")
    ap.add_argument("--scale", type=float, default=1.0, help="dev runs: multiply every config's stream count")
    ap.add_argument("--trace-out", default="", help="dev: save the per-unit (t_start, t_end, in_len) stamps of every config "
                                                     "as <prefix><config>.npz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-target-s", type=float, default=12.0)
    ap.add_argument("--side-cpu-target-s", type=float, default=3.0)
    ap.add_argument("--corpus-cache", default="", help="dev (profiling passes): keep generated corpora in this directory and reuse them")
    ap.add_argument("--allow-xlz-so", action="store_true", help="dev: accept a library swapped in with XLZ_SO (recorded in the line)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where rank 0 writes the full record (every config's roofline.issue, host_to_host, sweep, containers, library)")
    args = ap.parse_args()
    if os.environ.get("XLZ_SO") and not args.allow_xlz_so:
        raise SystemExit("bench.py: XLZ_SO is set (%s): the numbers would belong to another library than the tree's; "
                         "pass --allow-xlz-so for an A/B run" % os.environ["XLZ_SO"])

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
    head = args.headline
    side, extras = [], []
    if world == 1:  # the side configs, the sweep and the host-to-host legs are single-GPU measurements
        side = [] if args.configs == "none" else SIDE if args.configs == "all" else [c for c in args.configs.split(",") if c]
        side = [c for c in side if c != head]
        extras = [] if args.extras == "none" else EXTRAS if args.extras == "all" else [e for e in args.extras.split(",") if e]
    if ("h2h" in extras or "sweep" in extras) and "cfg2-T" not in side and head != "cfg2-T":
        side.append("cfg2-T")   # (their corpus)
    names = [head] + side

    # ---- synthetic corpora.  The pool's workers are ALL forked here, before anything touches the GPU (a process that has
    # initialised HIP must not fork); jobs can be handed to them at any time afterwards.
    ncpu = effective_cpus()
    workers = max(1, min((os.cpu_count() or 1) // max(1, min(world, 8)), 64))
    pool = ProcessPoolExecutor(max_workers=workers)
    pids = set(pool.map(_warm, range(workers)))   # the first submit forks the workers (fork context: all of them at once)
    log("[rank %d] %d corpus workers forked (%d host CPUs usable)" % (rank, len(pids), ncpu))
    corp, pending, wait_s = {}, {}, {}

    def cache_path(name):
        return os.path.join(args.corpus_cache, "xlz_corpus_%s_%g_%d_%d.pkl" % (name, args.scale, world, rank)) if args.corpus_cache else ""

    def start(name):
        if name in corp or name in pending:
            return
        spec = specs[name]
        cache = cache_path(name)
        if cache and os.path.exists(cache):
            import pickle
            corp[name] = pickle.load(open(cache, "rb"))
            wait_s[name] = 0.0
        elif name == head:
            # strong scaling: ONE batch (seed 1), this rank's shard of it (N = 1: all of it)
            weights = [out_size_of(spec)] * spec["streams"]
            pending[name] = start_corpus(pool, spec, 1, multigpu.partition_by_weight(weights, world)[rank])
        else:
            pending[name] = start_corpus(pool, spec, 1)

    def get(name):
        """the corpus, waiting for its jobs if they are still running"""
        if name not in corp:
            start(name)
        if name not in corp:   # (start() may have found it in the cache)
            t0 = time.time()
            corp[name] = pending.pop(name).result()
            wait_s[name] = round(time.time() - t0, 1)
            cache = cache_path(name)
            if cache:
                import pickle
                os.makedirs(args.corpus_cache, exist_ok=True)
                pickle.dump(corp[name], open(cache, "wb"), protocol=4)
            comp = corp[name][0]
            log("[rank %d] corpus %s: %d streams x %d B, ratio %.3f, waited %.1f s for it (%d workers)"
                % (rank, name, len(comp), out_size_of(specs[name]), sum(map(len, comp)) / (len(comp) * out_size_of(specs[name])),
                   wait_s[name], workers))
        return corp[name]

    def drain():
        """wait until the pool is idle: nothing compresses inside the headline's timed region or a host-timed leg"""
        for name in list(pending):
            get(name)
        if xz_pending[0] is not None:
            xz_get()

    xz_pending, xz_done = [None], [None]

    def xz_start():
        if "xz" in extras and xz_pending[0] is None and xz_done[0] is None:
            blocks = max(8, int(1024 * args.scale))
            xz_pending[0] = pool.map(_gen_xz_block, [(9000 + i, 1 << 20) for i in range(blocks)], chunksize=max(1, blocks // 256))

    def xz_get():
        if xz_done[0] is None and xz_pending[0] is not None:
            res = list(xz_pending[0])
            xz_pending[0] = None
            h = hashlib.sha256()
            for _, pl in res:
                h.update(pl)
            xz_done[0] = (b"".join(c for c, _ in res), h.digest())
        return xz_done[0]

    t_gen0 = time.time()
    get(head)
    gen_head_s = time.time() - t_gen0
    for name in side[:2]:   # compressed while torch is imported, HIP initialised and the headline uploaded
        start(name)

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
    lib_info = library()
    if not lib_info["built_from_tree"] and not args.allow_xlz_so:
        raise SystemExit("bench.py: %s was not built from the sources in the tree (build id %s, tree %s)"
                         % (lib_info["path"], lib_info["build_id"], lib_info["tree_source_id"]))
    ctx = lzma_amd.Context(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        multigpu.barrier(dist, torch.cuda.synchronize)

    def lone_leg(name):
        """live: bytes per second of ONE wave on this config's data (latency bound of roofline.issue)"""
        spec = specs[name]
        sample, osz = lone_wave_sample(spec, corp[name][0])
        b = make_batch(lzma_amd, ctx, dict(spec, segments=min(64, spec.get("segments", 1))), sample)
        b.run()
        b.sync()
        b.run()
        b.sync()
        t0, t1, _ = b.unit_trace()
        res = b.results()
        b.close()
        import numpy as np
        dur = (t1.astype(np.int64) - t0.astype(np.int64)) / 1e8  # s
        per_unit = sum(r[0] for r in res) / max(1, len(dur))
        return {"bytes_per_s_per_wave": round(float(per_unit / np.median(dur)), 1), "units": int(len(dur)),
                "unit_ms_median": round(float(np.median(dur)) * 1e3, 3),
                "source": "live: a launch of the config's first %d units alone (one wave each, the rest of the chip idle)" % len(dur)}

    def gpu_leg(name, steps, warmup, with_lone=True, quiet_host=False):
        spec = specs[name]
        comp, dig = get(name)
        osz = out_size_of(spec)
        t0 = time.time()
        batch = make_batch(lzma_amd, ctx, spec, comp)
        log("[rank %d] %s: batch created + uploaded in %.1f s" % (rank, name, time.time() - t0))
        if quiet_host:
            drain()
        t_local, kernel_ms, each = timed_steps(ctx, batch, steps, warmup, torch.cuda.synchronize, barrier if world > 1 else None)
        verify_all(batch, len(comp), osz, dig, "rank %d %s" % (rank, name))
        cin, cout, units = batch.stats()
        occ = occupancy(batch, kernel_ms)
        if args.trace_out:
            import numpy as np
            t0_, t1_, il_ = batch.unit_trace()
            np.savez_compressed(args.trace_out + name + ".npz", t_start=t0_, t_end=t1_, in_len=il_)
        batch.close()
        lone = lone_leg(name) if with_lone and "lone" in extras and spec["streams"] * spec.get("segments", 1) > 64 else None
        return t_local, kernel_ms, cin, cout, units, occ, lone, each

    def cpu_leg(name, target_s):
        if args.no_cpu_baseline or rank != 0 or world != 1:   # contract: rank 0 at N = 1 only
            return None
        spec = specs[name]
        comp, dig = corp[name]
        threads = int(os.environ.get("XLZ_BENCH_CPU_THREADS", "0")) or ncpu
        return cpu_baseline(spec, comp, dig, threads, target_s)

    def h2h_leg(name, reps=3, repeat=1):
        """xlz_decode_batch: host buffers in, host buffers out (PCIe included) -- the drop-in caller's rate.
        repeat > 1: the config's streams `repeat` times over in ONE call (a call of several wave rounds: the library
        pipelines upload / decode / download of sub-batches)"""
        import numpy as np
        from lzma_amd import _native as N
        spec = specs[name]
        comp, dig = corp[name]
        comp, dig = comp * repeat, dig * repeat
        n, osz = len(comp), out_size_of(spec)
        ins = [np.frombuffer(c, dtype=np.uint8) for c in comp]
        out = np.zeros((n, osz), dtype=np.uint8)   # one host buffer per stream, touched
        descs = (N.StreamDesc * n)()
        for i in range(n):
            descs[i].inp, descs[i].in_len = ins[i].ctypes.data, ins[i].size
            descs[i].out, descs[i].out_cap = out[i].ctypes.data, osz
            descs[i].format = lzma_amd.FMT_LZMA2_RAW if spec["fmt"] == "lzma2" else lzma_amd.FMT_LZMA_ALONE
            descs[i].dict_size = spec["dict"] if spec["fmt"] == "lzma2" else 0
        res = (N.Result * n)()
        calls = []
        for k in range(reps + 1):   # the first call allocates the pinned pools: not counted
            t0 = time.perf_counter()
            st = N.lib().xlz_decode_batch(ctx._h, descs, n, res)
            dt = time.perf_counter() - t0
            if st != 0:
                raise SystemExit("xlz_decode_batch failed: %d" % st)
            if k:
                calls.append((dt, ctx.last_call_stats()))
        calls.sort(key=lambda x: x[0])
        best = calls[(len(calls) - 1) // 2]   # the MEDIAN call (SURVEY 8d's timing protocol; the lower one of an even count)
        bad = [i for i in range(n) if res[i].status != 0 or res[i].out_len != osz]
        if bad:
            raise SystemExit("host_to_host %s: %d streams failed" % (name, len(bad)))
        with ThreadPoolExecutor(max_workers=16) as ex:
            ok = list(ex.map(lambda i: hashlib.sha256(out[i]).digest() == dig[i], range(n)))
        if not all(ok):
            raise SystemExit("host_to_host %s: decoded bytes differ from the plaintext" % name)
        dt, cs = best
        return {"name": name if repeat == 1 else "%s x%d" % (name, repeat), "value": round(n * osz / GIB / dt, 4), "unit": "GiB/s",
                "ms_per_call": round(dt * 1e3, 3),
                "streams": n, "bytes_per_stream": osz, "bit_exact": "all", "calls": reps + 1,
                "reported": "median of the %d calls after the first" % reps, "best_call": round(n * osz / GIB / calls[0][0], 4),
                "sub_batches": cs["sub_batches"], "slices": cs["slices"],
                "phases_ms": {"pack_upload": round(cs["upload_ms"], 3), "decode": round(cs["decode_ms"], 3),
                              "download_scatter": round(cs["download_ms"], 3),
                              "note": "sub_batches > 1: a pipeline -- pack_upload = until the first sub-batch was on the device, decode = "
                                      "first launch to last results (the other uploads and downloads run inside it), "
                                      "download_scatter = what was left after that; slices > 1: a call of one wave round as a "
                                      "sequence of launches -- pack_upload = until the first launch could start (the heads of the "
                                      "inputs), decode = the launches (HIP events; the tails' upload and the downloads of all "
                                      "slices but the last run inside it), download_scatter = what was left after them"},
                "slot_occupancy": round(cs["slot_occupancy"], 4), "wave_slots": cs["wave_slots"],
                "path": "xlz_decode_batch: pageable host buffers in, pageable host buffers out, through the library's pinned pools"}

    # ------------------------------------------------------------ the headline: one workload at every N ----
    spec = specs[head]
    t_head0 = time.time()
    t_local, kernel_ms, cin, cout, units, occ, lone, each = gpu_leg(head, args.steps, args.warmup, quiet_host=True)
    for name in side[2:]:   # the rest of the corpora: compressed while the side legs run (their launches are timed by HIP events)
        start(name)
    xz_start()
    t_max = multigpu.max_over_ranks(t_local, dist, device=red_dev)
    n_max = multigpu.max_over_ranks(float(len(corp[head][0])), dist, device=red_dev)
    sums = [float(len(corp[head][0])), float(cin), float(cout)]
    if world > 1:
        t = torch.tensor(sums, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t)  # 24 bytes of bookkeeping, not data: streams / bytes over all shards
        sums = [float(x) for x in t.tolist()]
    total_streams = int(sums[0])
    assert total_streams == spec["streams"], "the shards do not add up to the batch"
    total_out = spec["streams"] * out_size_of(spec) * args.steps
    head_res = {
        "value": round(total_out / GIB / t_max, 4), "ms_per_step": round(t_max / args.steps * 1e3, 3),
        "ms_per_step_median": round(median(each), 3) if each else None,
        "roofline": roofline(head, cin, cout, units, kernel_ms, occ, lone),
    }
    if each:
        head_res["roofline"]["kernel_ms_median"] = round(median(each), 3)
        head_res["roofline"]["kernel_ms_steps"] = [round(x, 3) for x in each]
    log("[headline] %s: value %.4f GiB/s, ms_per_step %.3f (median step %s ms), kernel_ms %.3f, roofline.frac %.6f"
        % (head, head_res["value"], head_res["ms_per_step"], head_res["ms_per_step_median"], kernel_ms, head_res["roofline"]["frac"]))
    if world > 1:
        # achieved / kernel_ms / units are rank 0's shard; the committed PMC passes profiled the WHOLE batch on one GPU, so
        # their byte counts do not describe this launch: traffic is scaled by the shard's share of the algorithmic bytes
        # and labelled, the raw whole-batch figures stay under traffic_from_profile
        rf = head_res["roofline"]
        rf["note"] = "rank 0's shard"
        if rf.get("traffic") and sums[1] + sums[2] > 0:
            share = (cin + cout) / (sums[1] + sums[2])
            rf["traffic"] = round(rf["traffic"] * share, 1)
            rf["traffic_from_profile"]["scaled_by"] = round(share, 6)
            rf["traffic_from_profile"]["note"] = ("profiled on the whole batch at N = 1; traffic = that figure x this shard's "
                                                  "share of the algorithmic bytes")
    head_leg_s = time.time() - t_head0

    # ------------------------------------------------------------ N = 1: what the N-GPU points of this curve would be ----
    # The path has no inter-GPU traffic (the batch is split by stream, every rank decodes its shard on its own), so the
    # time of rank 0's shard on ONE GPU is the time of the N-GPU step -- as far as one GPU can tell: a projection, labelled so.
    projection = None
    if world == 1 and "shard" in extras:
        comp, dig = corp[head]
        weights = [out_size_of(spec)] * spec["streams"]
        projection = []
        for nv in SHARD_COUNTS:
            idx = multigpu.partition_by_weight(weights, nv)[0]
            b = make_batch(lzma_amd, ctx, spec, [comp[i] for i in idx])
            tl, kms, ea = timed_steps(ctx, b, 5, 1, torch.cuda.synchronize)
            verify_all(b, len(idx), out_size_of(spec), [dig[i] for i in idx], "shard 0 of %d" % nv)
            wg, _ = b.launch_info()
            b.close()
            # the MEDIAN step by HIP events (five steps of 30-120 ms: one host hiccup moves their wall-clock mean by 15 %);
            # the wall-clock mean is kept beside it
            step_ms = median(ea) if ea else tl / 5 * 1e3
            v = spec["streams"] * out_size_of(spec) / GIB / (step_ms / 1e3)
            projection.append({"n_gpus": nv, "streams_on_rank_0": len(idx), "wave_slots": wg, "rounds": round(len(idx) / max(1, wg), 2),
                               "ms_per_step": round(step_ms, 3), "ms_per_step_wall_mean": round(tl / 5 * 1e3, 3),
                               "steps_ms": [round(x, 3) for x in ea],
                               "value_projected": round(v, 3), "efficiency_projected": round(v / (nv * head_res["value"]), 4)})
        log("[projection] " + ", ".join("N=%d: %.1f GiB/s (%.2f of linear; %d streams = %.2f rounds on %d slots)" % (
            x["n_gpus"], x["value_projected"], x["efficiency_projected"], x["streams_on_rank_0"], x["rounds"], x["wave_slots"]) for x in projection))

    # ------------------------------------------------------------ N = 1: the other configs and the extras ----
    results = {}
    t_side0 = time.time()
    for name in side:
        sp = specs[name]
        steps, warmup = args.side_steps, 1
        tl, kms, ci, co, un, oc, ln, ea = gpu_leg(name, steps, warmup)
        n = len(corp[name][0])
        results[name] = {
            "name": name, "baseline_config": sp["baseline"], "workload": workload_text(name, sp),
            # the HIP-event time of the launches: corpora of later configs are being compressed on the host meanwhile
            "value": round(n * out_size_of(sp) / GIB / (kms / 1e3), 4), "unit": "GiB/s", "steps": steps, "warmup": warmup,
            "value_wall": round(n * out_size_of(sp) * steps / GIB / tl, 4),   # (rounds <= 3 reported this one as `value`)
            "ms_per_step": round(kms, 3), "ms_per_step_wall": round(tl / steps * 1e3, 3), "kernel_ms": round(kms, 3),
            "kernel_ms_median": round(median(ea), 3) if ea else None,
            "timed_with": "HIP events on the kernel's stream around the timed steps",
            "streams": n, "bytes_per_stream": out_size_of(sp), "compression_ratio": round(ci / co, 4),
            "bit_exact": "all", "roofline": roofline(name, ci, co, un, kms, oc, ln),
        }
        log("[side] %s: %.4f GiB/s, kernel_ms %.3f, roofline.frac %.6f" % (name, results[name]["value"], kms, results[name]["roofline"]["frac"]))
    side_legs_s = time.time() - t_side0
    drain()    # from here on the host is quiet: the host-timed legs and the CPU baselines
    pool.shutdown()
    gen_s = time.time() - t_gen0
    xz = xz_done[0]
    t_extra0 = time.time()
    h2h, sweep, containers = None, None, None
    if "h2h" in extras:
        h2h = [h2h_leg(n) for n in dict.fromkeys([head, "cfg2-T"]) if n in corp]
        if "cfg2-T" in corp and args.scale == 1.0:
            h2h.append(h2h_leg("cfg2-T", reps=2, repeat=4))   # 16 384 x 1 MiB: four wave rounds in one call
        log("[h2h] " + ", ".join("%s %.2f GiB/s" % (x["name"], x["value"]) for x in h2h))
    if "sweep" in extras and "cfg2-T" in corp:
        sweep = []
        comp, dig = corp["cfg2-T"]
        sp = specs["cfg2-T"]
        for k in SWEEP_COUNTS:
            if k >= len(comp):
                continue
            b = make_batch(lzma_amd, ctx, sp, comp[:k])
            tl, kms, _ = timed_steps(ctx, b, 2, 1, torch.cuda.synchronize)
            oc = occupancy(b, kms)
            verify_all(b, k, sp["size"], dig[:k], "sweep %d" % k)
            b.close()
            sweep.append({"streams": k, "value": round(k * sp["size"] * 2 / GIB / tl, 4), "unit": "GiB/s", "kernel_ms": round(kms, 3),
                          "slot_occupancy": oc and oc["slot_occupancy"], "bit_exact": "all"})
    if "xz" in extras and xz is not None:
        import numpy as np
        data, want = xz
        _, total = lzma_amd.xz_index(data)
        out = np.zeros(total, dtype=np.uint8)   # the caller's buffer, touched (as in host_to_host)
        times = []
        for _ in range(3):   # the first call allocates the pinned pools
            t0 = time.perf_counter()
            n_out = lzma_amd.xz_decode_into(ctx, data, out, verify=True)
            times.append((time.perf_counter() - t0, ctx.last_call_stats()))
        dt, cs = min(times, key=lambda x: x[0])
        cs_xz = {"pack_upload": round(cs["upload_ms"], 3), "decode": round(cs["decode_ms"], 3),
                 "download_scatter": round(cs["download_ms"], 3),
                 "index_and_check": round(dt * 1e3 - cs["upload_ms"] - cs["decode_ms"] - cs["download_ms"], 3),
                 "note": "the three phases of the xlz_decode_batch call inside; index_and_check = the rest of the call (block "
                         "index, CRC64 of every block on host threads)"}
        out = out[:n_out]
        if hashlib.sha256(out).digest() != want:
            raise SystemExit("xz container: decoded bytes differ from the plaintext")
        containers = [{"name": "xz-blocks", "workload": "one .xz file of %d independently compressed 1 MiB blocks (LZMA2, 1 MiB "
                       "dictionary, CRC64), %s; xlz_xz_decode called with the caller's buffers: file in host memory -> decoded bytes in "
                       "host memory, every block's CRC64 verified on the host" % (max(8, int(1024 * args.scale)), ENC_FAST_NAME),
                       "value": round(len(out) / GIB / dt, 4), "unit": "GiB/s", "ms_per_call": round(dt * 1e3, 3),
                       "compressed_bytes": len(data), "decoded_bytes": len(out), "bit_exact": "all", "calls": 3, "reported": "best call", "phases_ms": cs_xz}]
        del out
    extras_s = time.time() - t_extra0

    # ---- CPU legs after all GPU work: the host cores are quiet
    t_cpu0 = time.time()
    cpu_head = cpu_leg(head, args.cpu_target_s)
    for name in side:
        results[name]["cpu_baseline"] = cpu_leg(name, args.side_cpu_target_s)
        log("[cpu] %s: %s" % (name, results[name]["cpu_baseline"] and results[name]["cpu_baseline"]["value"]))
    if sweep and not args.no_cpu_baseline:
        comp, dig = corp["cfg2-T"]
        threads = int(os.environ.get("XLZ_BENCH_CPU_THREADS", "0")) or ncpu
        for e in sweep:
            e["cpu_baseline"] = cpu_baseline(specs["cfg2-T"], comp[:e["streams"]], dig[:e["streams"]], threads, 2.0)
        full = results.get("cfg2-T")
        pts = [(e["streams"], e["value"], e["cpu_baseline"]["value"]) for e in sweep]
        if full and full.get("cpu_baseline"):
            pts.append((full["streams"], full["value"], full["cpu_baseline"]["value"]))
        wins = [k for k, g, c in pts if g >= c]
        sweep = {"workload": "the first k streams of cfg2-T (device-resident, kernel rate) against the CPU baseline on the same k "
                             "streams with %d host threads" % threads,
                 "points": sweep + ([{"streams": full["streams"], "value": full["value"], "unit": "GiB/s", "kernel_ms": full["kernel_ms"],
                                      "slot_occupancy": full["roofline"].get("issue", {}).get("slot_occupancy"),
                                      "cpu_baseline": full["cpu_baseline"]}] if full else []),
                 "break_even_streams": min(wins) if wins else None,
                 "note": "one wave decodes one stream (LZMA1 is a serial chain, decompress.go:13); the chip holds 4096 waves: below "
                         "the break-even the host's cores are the faster decoder (include/xlz.h, INTEGRATION.md)"}
    if containers and not args.no_cpu_baseline:
        import lzma
        # bounded sample: the first 128 blocks (each block is a stream of its own, so a prefix of the file is a file)
        pos, nblk = 0, 0
        dec = lzma.LZMADecompressor()
        t0 = time.perf_counter()
        n_out = 0
        while pos < len(xz[0]) and nblk < 128:
            n_out += len(dec.decompress(xz[0][pos:pos + (1 << 20)]))
            pos += 1 << 20
            if dec.eof:
                rest = dec.unused_data
                pos -= len(rest)
                dec = lzma.LZMADecompressor()
                nblk += 1
        dt = time.perf_counter() - t0
        containers[0]["cpu_baseline"] = {"value": round(n_out / GIB / dt, 4), "unit": "GiB/s", "cores": 1, "kind": "liblzma 5.x (xz) via Python",
                                         "sample": "the first %d blocks (%d MiB decoded) in %.2f s on one thread (xz 5.2 decodes a file on one "
                                                   "thread; the reference has no container code)" % (nblk, n_out >> 20, dt)}
    sanity = None
    if cpu_head is not None and spec["fmt"] == "lzma1":
        try:
            sanity = liblzma_sanity(corp[head][0], out_size_of(spec), ncpu)
        except Exception as e:  # a sanity line must never fail the bench
            log("liblzma sanity line skipped: %r" % (e,))
    cpu_legs_s = time.time() - t_cpu0
    if world > 1:
        barrier()
    if rank == 0:
        full = {
            "metric": "decompressed GiB/s (aggregate batch)", "value": head_res["value"], "unit": "GiB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head_res["ms_per_step"],
            "ms_per_step_median": head_res["ms_per_step_median"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_text(head, spec, per_gpu=False) + "; ONE seeded batch split by stream over %d GPU%s"
                       % (world, "" if world == 1 else "s"),
                       "streams_total": total_streams, "streams_largest_shard": int(n_max),
                       "bytes_per_stream": out_size_of(spec), "compression_ratio": round(sums[1] / sums[2], 4), "bit_exact": "all",
                       "parallelism": "one batch sharded by stream x%d (partition_by_weight), no collective" % world,
                       "corpus_generation_s": round(gen_s, 1), "kernel_rev": lib_info["kernel_id"], "library": lib_info},
            "roofline": head_res["roofline"],
            "cpu_baseline": cpu_head,
            "timing": {"total_s": round(time.time() - t_main0, 1), "headline_corpus_s": round(gen_head_s, 1),
                       "waited_for_corpora_s": wait_s, "headline_leg_s": round(head_leg_s, 1), "side_legs_s": round(side_legs_s, 1),
                       "extras_s": round(extras_s, 1), "cpu_legs_s": round(cpu_legs_s, 1),
                       "note": "side corpora are compressed by the worker pool while the GPU legs run (never inside the headline's timed "
                               "region, a host-timed leg or a CPU baseline)"},
        }
        if world == 1:
            full["cpu_sanity_liblzma"] = sanity
            full["host_to_host"] = h2h
            full["stream_count_sweep"] = sweep
            full["containers"] = containers
            full["configs"] = [results[n] for n in side if n in results]
            if projection:
                full["scaling_projection"] = {
                    "label": "projection: one GPU, rank 0's 1/N shard of the same batch, no inter-GPU traffic in this path",
                    "n1_value": head_res["value"], "points": projection,
                    "note": "value_projected = bytes of the WHOLE batch / the shard's median step (HIP events) on one GPU; efficiency_projected = that / "
                            "(N x the N = 1 value).  What it cannot see: N processes sharing one host's PCIe root and cores"}
        written = write_detail(full, args.detail_out)
        line = compact_line(full, os.path.relpath(written[0], ROOT) if written else None)
        log("[bench] value %.4f %s  ms_per_step %.3f  n_gpus %d  roofline.frac %.6f  cpu_baseline %s  (%.0f s; detail: %s)"
            % (line["value"], line["unit"], line["ms_per_step"], world, line["roofline"]["frac"],
               line["cpu_baseline"] and line["cpu_baseline"]["value"], time.time() - t_main0, ", ".join(written) or "not written"))
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
