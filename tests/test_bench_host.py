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


def test_the_headline_is_one_workload_at_every_n_and_the_xz_file_is_what_liblzma_reads():
    """the curve's points must be one workload: cfg3 is the default headline, the side list does not
    contain it, every corpus uses the one encoder setting except the explicitly named preset-6 entry"""
    assert bench.HEADLINES[0] == "cfg3" and "cfg3" not in bench.SIDE
    assert all(bench.CONFIGS[n]["enc"] == bench.ENC_FAST for n in bench.CONFIGS if n != "cfg2-T-p6")
    with ProcessPoolExecutor(max_workers=2) as pool:
        data, want = bench.xz_file(pool, 5, 30_000)
    p = lzma.decompress(data)
    assert len(p) == 150_000 and hashlib.sha256(p).digest() == want
    import lzma_amd
    blocks, total = lzma_amd.xz_index(data)
    assert len(blocks) == 5 and total == 150_000
    # the lone-wave sample of an LZMA2 config is a valid stream of its first segments
    spec = _tiny("cfg4", segments=70, size=5_000)
    with ProcessPoolExecutor(max_workers=2) as pool:
        comp, dig = bench.make_corpus(pool, spec, 1)
    sample, osz = bench.lone_wave_sample(spec, comp)
    assert osz == 64 * 5_000
    got = oracle.lzma2_raw(sample[0], spec["dict"], osz)
    assert got[1] == 0 and len(got[0]) == osz


def test_issue_bounds_arithmetic():
    prof = {"issue": {"salu_per_decoded_byte": 20.0, "valu_per_decoded_byte": 20.0, "source": "x"}}
    r = bench.issue_bounds(prof, 4 << 30, 200.0, {"bytes_per_s_per_wave": 6e6}, 4096)
    # scalar port binds: 256 CUs x 2.4 GHz / (20 / 0.97) cycles per byte
    assert abs(r["issue_bound_GBps"] - 256 * 2.4 / (20 / 0.97)) < 0.01
    assert abs(r["latency_bound_GBps"] - 6e6 * 4096 / 1e9) < 0.01
    assert r["binding"] == "latency" and 0 < r["frac_of_bound"] < 1
    assert bench.issue_bounds(None, 1 << 30, 100.0, None, None) == {"achieved_decoded_GBps": round((1 << 30) / 0.1 / 1e9, 3)}
    # a launch of stored chunks (a plain copy): the HBM roof in decoded bytes (8 TB/s x decoded / algorithmic) is what binds
    r = bench.issue_bounds({"issue": {"salu_per_decoded_byte": 0.005, "valu_per_decoded_byte": 0.02}}, 1 << 30, 0.53,
                           {"bytes_per_s_per_wave": 9e9}, 4096, 2 * (1 << 30) + 1000)
    assert r["binding"] == "hbm" and abs(r["hbm_bound_GBps"] - 4000.0) < 1 and 0.4 < r["frac_of_bound"] < 0.6
    assert r["bounds_known"] == ["issue", "latency", "hbm"]
