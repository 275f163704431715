#!/usr/bin/env python3
"""bench.py -- batched LZMA decode throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (the gfx950 decode kernel) over one batch of synthetic
compressed streams that is already resident in HBM.  At N = 1 the workload is BASELINE.json
configs[1]: 4096 independent LZMA1 streams, lc=3/lp=0/pb=2, 64 KiB dictionary, 1 MiB
uncompressed each.  For N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank decodes its own batch of that shape: streams are independent, so the batch is sharded by
stream with no data-path collective ("scaling": "weak"); torch.distributed only provides the
barrier and the max-over-ranks of the timed region.

Rank 0 prints ONE JSON line.  `value` = decompressed GiB of all ranks / max-over-ranks time.
`roofline` prices the decode kernel against HBM bandwidth with algorithmic bytes (compressed
bytes read once + decoded bytes written once), timed with HIP events on the kernel's own
stream.  `cpu_baseline` times the CPU oracle (a C restatement of the Go reference's algorithm;
the Go toolchain does not exist here) on a bounded sample of the same streams.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md chip table)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--size", type=int, default=1 << 20, help="uncompressed bytes per stream")
    ap.add_argument("--family", default="T", choices=["T", "R", "M", "Z"], help="plaintext family (corpus.py)")
    ap.add_argument("--dict", type=int, default=65536)
    ap.add_argument("--lc", type=int, default=3)
    ap.add_argument("--lp", type=int, default=0)
    ap.add_argument("--pb", type=int, default=2)
    ap.add_argument("--preset", type=int, default=6)
    ap.add_argument("--distinct", type=int, default=0,
                    help="generate only this many distinct streams and reuse them (dev runs; 0 = all distinct)")
    ap.add_argument("--format", default="lzma1", choices=["lzma1", "lzma2"],
                    help="lzma2: every stream is ONE raw LZMA2 stream made of --segments independently compressed "
                         "segments of --size bytes (BASELINE config 4: chunk-parallel units)")
    ap.add_argument("--segments", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-target-s", type=float, default=15.0)
    ap.add_argument("--verify", default="all", choices=["all", "sample", "none"])
    args = ap.parse_args()

    from lzma_amd import multigpu
    rank, world, local_rank = multigpu.env_rank()
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))

    # ---- synthetic corpus: every rank its own seeds.  Generated BEFORE anything touches the
    # GPU: the generator forks worker processes.
    import corpus
    t0 = time.time()
    nd = args.distinct if args.distinct > 0 else args.streams
    ncpu = effective_cpus()
    workers = max(1, min((os.cpu_count() or 1) // max(1, min(world, 8)), 64))
    out_size = args.size * (args.segments if args.format == "lzma2" else 1)  # decoded bytes per stream
    if args.format == "lzma2":
        comp, digests = corpus.make_lzma2_batch(args.family, nd, args.segments, args.size,
                                                base_seed=1 + rank * 1_000_003, workers=workers, dict_size=args.dict,
                                                lc=args.lc, lp=args.lp, pb=args.pb, preset=args.preset)
    else:
        comp, digests = corpus.make_alone_batch(args.family, nd, args.size, base_seed=1 + rank * 1_000_003,
                                                workers=workers, dict_size=args.dict, lc=args.lc, lp=args.lp,
                                                pb=args.pb, preset=args.preset)
    gen_s = time.time() - t0
    comp_bytes = sum(len(comp[i % nd]) for i in range(args.streams))
    log("[rank %d] corpus: %d streams (%d distinct) x %d B, ratio %.3f, generated in %.1f s with %d workers"
        % (rank, args.streams, nd, out_size, comp_bytes / (args.streams * out_size), gen_s, workers))

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

    # ---- upload once: inputs resident in HBM before the timed region -------------------
    ctx = lzma_amd.Context(local_rank)
    t0 = time.time()
    if args.format == "lzma2":
        batch = lzma_amd.Batch(ctx, [lzma_amd.Stream(comp[i % nd], lzma_amd.FMT_LZMA2_RAW, out_cap=out_size,
                                                     dict_size=args.dict) for i in range(args.streams)])
    else:
        batch = lzma_amd.Batch(ctx, [lzma_amd.Stream(comp[i % nd], out_cap=out_size) for i in range(args.streams)])
    log("[rank %d] batch created + uploaded in %.1f s" % (rank, time.time() - t0))

    def barrier():
        multigpu.barrier(dist, torch.cuda.synchronize)

    for _ in range(args.warmup):
        batch.run()
    batch.sync()

    barrier()
    t_start = time.perf_counter()
    ctx.event_record(0)
    for _ in range(args.steps):
        batch.run()
    ctx.event_record(1)
    batch.sync()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t_start
    barrier()
    kernel_ms = ctx.event_elapsed_ms(0, 1) / max(1, args.steps)  # HIP events on the kernel's stream

    t_max = multigpu.max_over_ranks(t_local, dist, device="cuda" if backend == "nccl" else "cpu")

    # ---- verification, outside the timed region -----------------------------------------
    res = batch.results()
    bad = [i for i in range(args.streams) if res[i][1] != 0 or res[i][0] != out_size]
    if bad:
        raise SystemExit("rank %d: %d streams failed to decode (first: %d, status %d, out_len %d)"
                         % (rank, len(bad), bad[0], res[bad[0]][1], res[bad[0]][0]))
    cin, cout, units = batch.stats()
    if args.verify != "none":
        idx = list(range(args.streams)) if args.verify == "all" else list(range(0, args.streams, max(1, args.streams // 64)))
        t0 = time.time()

        def check(i):
            return hashlib.sha256(batch.download(i, out_size)).digest() == digests[i % nd]
        with ThreadPoolExecutor(max_workers=16) as ex:
            ok = list(ex.map(check, idx))
        if not all(ok):
            raise SystemExit("rank %d: decoded bytes differ from the plaintext for %d streams" % (rank, ok.count(False)))
        log("[rank %d] verified %d streams bit-exact (sha256 of every output byte) in %.1f s"
            % (rank, len(idx), time.time() - t0))

    # ---- CPU baseline: rank 0, N = 1 only --------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        fmt_code = 2 if args.format == "lzma2" else 0
        import oracle
        threads = int(os.environ.get("XLZ_BENCH_CPU_THREADS", "0")) or ncpu
        # calibrate on one stream per thread, then size the sample for ~cpu_target_s
        k = min(threads, args.streams)
        _, _, per_round = oracle.decode_batch_mt([comp[i % nd] for i in range(k)], [out_size] * k, threads,
                                                 fmt=fmt_code, dict_size=args.dict, timing=True)
        rounds = max(1, int(args.cpu_target_s / max(per_round, 1e-3)))
        n_sample = min(args.streams, max(k, rounds * k))
        sample = [comp[i % nd] for i in range(n_sample)]
        outs, sts, dt = oracle.decode_batch_mt(sample, [out_size] * n_sample, threads, fmt=fmt_code,
                                               dict_size=args.dict, timing=True)
        assert all(s == 0 for s in sts)
        for i in range(0, n_sample, max(1, n_sample // 16)):
            buf, n_out = outs[i]
            assert hashlib.sha256(buf.raw[:n_out]).digest() == digests[i % nd]
        cpu = {"value": round(n_sample * out_size / GIB / dt, 4), "unit": "GiB/s", "cores": threads, "kind": "port",
               "sample": "%d of the %d streams (%d MiB decoded) in %.2f s of decode wall time; C restatement of the "
                         "Go reference's algorithm (oracle/xlz_oracle.c, gcc -O2), one stream per thread on all "
                         "%d host CPUs; Go toolchain absent" % (n_sample, args.streams, n_sample * args.size >> 20,
                                                                dt, threads)}

    if rank == 0:
        total_out = world * args.streams * out_size * args.steps
        value = total_out / GIB / t_max
        algo_bytes = cin + cout  # per launch: compressed bytes read once + decoded bytes written once
        achieved = algo_bytes / 1e9 / (kernel_ms / 1e3)
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes over this same
        # command (tools/profile_bench.sh); bench.py cannot profile itself, so it reports the committed
        # measurement when it was taken on the same workload, else null.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_default.json")))
            workload_now = ("%d independent LZMA1 (.lzma) streams per GPU" % args.streams if args.format == "lzma1" else "")
            if args.format == "lzma1" and tj["workload"].startswith(workload_now) and \
                    ("%d B uncompressed per stream, family %s" % (out_size, args.family)) in tj["workload"]:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "decompressed GiB/s (aggregate batch)",
            "value": round(value, 4),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": ("%d independent LZMA1 (.lzma) streams per GPU" % args.streams if args.format == "lzma1" else
                             "%d raw LZMA2 stream(s) per GPU of %d dictionary-reset segments each" % (args.streams,
                                                                                                     args.segments))
                            + ", lc=%d lp=%d pb=%d, %d KiB dict, %d B uncompressed per stream, family %s (corpus.py), "
                            "liblzma preset %d, inputs resident in HBM" % (args.lc, args.lp, args.pb, args.dict >> 10,
                                                                           out_size, args.family, args.preset),
                "streams_per_gpu": args.streams, "bytes_per_stream": out_size,
                "compression_ratio": round(comp_bytes / (args.streams * out_size), 4),
                "bit_exact": args.verify, "parallelism": "shard-by-stream x%d, no collective" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "kernel": "xlz::xlz_decode_kernel", "kernel_ms": round(kernel_ms, 3),
                "algorithmic_bytes_per_launch": algo_bytes, "units_per_launch": units,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)

    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
