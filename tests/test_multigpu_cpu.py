"""CPU: the N>1 path (shard by stream, no data collective) with world_size-2 gloo processes
and with the in-process multi-device driver.  The decode itself is stood in for by the CPU
oracle here (this container has no GPU); the GPU path is covered by tests marked gpu."""
import hashlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import corpus
import oracle
from lzma_amd import multigpu


def test_rank_shard_covers_everything():
    for n in (0, 1, 7, 4096, 65537):
        for world in (1, 2, 3, 8):
            spans = [multigpu.rank_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_partition_by_weight_is_balanced_and_complete():
    w = [1 << 20] * 100 + [5 << 20] * 7 + [123] * 50
    for k in (1, 2, 4, 8):
        shards = multigpu.partition_by_weight(w, k)
        assert sorted(i for s in shards for i in s) == list(range(len(w)))
        loads = [sum(w[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(w)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, w, _ = multigpu.env_rank()
    assert (r, w) == (rank, world)
    lo, hi = multigpu.rank_shard(n_streams, r, w)
    # every rank builds the same seeded corpus description and decodes only its slice
    digests = []
    multigpu.barrier(dist)
    for i in range(lo, hi):
        p = corpus.plain("TRMZ"[i % 4], 1000 + i, 20_000)
        c = corpus.compress_alone(p, preset=1)
        out, st, _ = oracle.lzma1_alone(c, len(p))
        assert st == 0 and out == p
        digests.append((i, hashlib.sha256(out).hexdigest()))
    multigpu.barrier(dist)
    t = multigpu.max_over_ranks(1.0 + rank, dist)      # MAX over ranks, as bench.py takes it
    assert t == float(world)
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)           # control-plane only: 64-byte digests
    if rank == 0:
        q.put(sorted(x for part in gathered for x in part))
    dist.destroy_process_group()


def test_world_size_2_gloo_shards_without_data_collective():
    n = 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [i for i, _ in got] == list(range(n))
    for i, h in got:
        assert h == hashlib.sha256(corpus.plain("TRMZ"[i % 4], 1000 + i, 20_000)).hexdigest()


def test_in_process_multi_device_driver_keeps_order():
    import lzma_amd
    ps = [corpus.plain("T", 2000 + i, 5_000 + 100 * i) for i in range(9)]
    streams = [lzma_amd.Stream(corpus.compress_alone(p, preset=1), out_cap=len(p)) for p in ps]
    seen = {}

    def fake_decode(device, shard):  # stands in for the HIP path on a GPU-less machine
        seen[device] = len(shard)
        return [oracle.lzma1_alone(s.data, s.out_cap) for s in shard]
    res = multigpu.decode_batch_multi(streams, [0, 1, 2], decode=fake_decode)
    assert [r[0] for r in res] == ps and all(r[1] == 0 for r in res)
    assert sum(seen.values()) == 9 and len(seen) == 3
