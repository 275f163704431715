"""Sharding of a batch of independent streams over the GPUs of one node.

Streams share no state (the reference's readers have no mutable globals), so the batch is
split by stream and every GPU decodes its shard on its own: no RCCL, no xGMI traffic.  Two
ways to drive it:

* one process per GPU (torch.distributed.run, what bench.py does): `rank_shard` gives each
  rank its contiguous slice; `max_over_ranks` / `barrier` are the only collectives and carry
  8 bytes of timing, not data;
* one process, one host thread per GPU: `decode_batch_multi`.
"""
import os
import threading


def env_rank():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def rank_shard(n_items, rank, world):
    """Contiguous, near-equal slice [lo, hi) of n_items for `rank` (equal-size items)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def partition_by_weight(weights, n_shards):
    """Longest-processing-time greedy: index lists with near-equal total weight (e.g. the
    uncompressed size of each stream).  Deterministic."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    shards = [[] for _ in range(n_shards)]
    loads = [0] * n_shards
    for i in order:
        k = min(range(n_shards), key=lambda s: (loads[s], s))
        shards[k].append(i)
        loads[k] += weights[i]
    for s in shards:
        s.sort()
    return shards


def barrier(dist=None, sync=None):
    """Barrier over ranks (if torch.distributed is initialised) + device sync (if given)."""
    if dist is not None and dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync is not None:
        sync()


def max_over_ranks(value, dist=None, device=None):
    """MAX of a python float over all ranks (the timed region's duration)."""
    if dist is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def decode_batch_multi(streams, devices, decode=None):
    """Decode `streams` on several GPUs from one process: one host thread and one context
    per device, shards balanced by compressed bytes, results returned in input order.

    `decode(device, shard_streams) -> list of results` defaults to the HIP path; tests pass a
    stand-in to exercise the sharding on machines without GPUs.
    """
    streams = list(streams)
    if decode is None:
        import lzma_amd

        def decode(device, shard):
            ctx = lzma_amd.Context(device)
            try:
                return lzma_amd.decode_batch(ctx, shard)
            finally:
                ctx.close()
    # balanced by COMPRESSED bytes, like the C entry (xlz_decode_batch_multi) and the kernel's own work queue: decode time
    # tracks the number of binary decisions, which tracks them
    shards = partition_by_weight([len(s.data) for s in streams], len(devices))
    results = [None] * len(streams)
    errors = []

    def work(dev, idx):
        try:
            out = decode(dev, [streams[i] for i in idx])
            for i, r in zip(idx, out):
                results[i] = r
        except Exception as e:  # surfaced to the caller below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(d, idx)) for d, idx in zip(devices, shards) if idx]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
