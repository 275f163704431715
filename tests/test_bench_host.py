"""CPU: the host-side legs of bench.py (corpus generation per config, the strong-scaling shard
recipe, the CPU-baseline sampler) at toy sizes.  The GPU legs run on the GPU box."""
import hashlib
import lzma
from concurrent.futures import ProcessPoolExecutor

import bench
import oracle
from lzma_amd import multigpu


def _tiny(name, **over):
    spec = dict(bench.CONFIGS[name])
    spec.update(over)
    return spec


def test_every_config_has_a_corpus_recipe_and_a_cpu_baseline():
    with ProcessPoolExecutor(max_workers=2) as pool:
        for name in bench.CONFIGS:
            if bench.CONFIGS[name]["fmt"] == "lzma2":
                spec = _tiny(name, segments=6, size=20_000)
            else:
                spec = _tiny(name, streams=5, size=min(bench.CONFIGS[name]["size"], 30_000))
            comp, dig = bench.make_corpus(pool, spec, 1)
            osz = bench.out_size_of(spec)
            assert len(comp) == spec["streams"] == len(dig)
            for c, d in zip(comp, dig):  # liblzma agrees with the digests; the oracle agrees with liblzma
                if spec["fmt"] == "lzma1":
                    p = lzma.decompress(c, format=lzma.FORMAT_ALONE)
                    got = oracle.lzma1_alone(c, osz)
                else:
                    p = lzma.decompress(c, format=lzma.FORMAT_RAW,
                                        filters=[{"id": lzma.FILTER_LZMA2, "dict_size": spec["dict"], "lc": spec["lc"],
                                                  "lp": spec["lp"], "pb": spec["pb"]}])
                    got = oracle.lzma2_raw(c, spec["dict"], osz)
                assert len(p) == osz and hashlib.sha256(p).digest() == d
                assert got[0] == p and got[1] == 0
            cpu = bench.cpu_baseline(spec, comp, dig, 2, 0.05)
            assert cpu["value"] > 0 and cpu["kind"] == "port" and cpu["cores"] >= 1
            assert name in bench.workload_text(name, spec)


def test_strong_scaling_shards_are_a_partition_of_the_one_batch():
    spec = _tiny("cfg3", streams=37, size=4000)
    with ProcessPoolExecutor(max_workers=2) as pool:
        whole, wd = bench.make_corpus(pool, spec, 1)
        seen = {}
        for world in (2, 4):
            shards = multigpu.partition_by_weight([bench.out_size_of(spec)] * spec["streams"], world)
            assert sorted(i for s in shards for i in s) == list(range(spec["streams"]))
            assert max(map(len, shards)) - min(map(len, shards)) <= 1
            for r, sh in enumerate(shards):
                comp, dig = bench.make_corpus(pool, spec, 1, sh)
                for i, c, d in zip(sh, comp, dig):  # stream i is the same bytes whichever rank makes it
                    assert c == whole[i] and d == wd[i]
                    seen[i] = True
        assert len(seen) == spec["streams"]
