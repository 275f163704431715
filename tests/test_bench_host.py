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


def _stub_roofline(name):
    """a roofline object as bench.roofline() builds it, every optional part present (the round-3 line carried all of
    this for eight configs and grew to 32 KB; the driver could not parse it)"""
    return {"bound": "hbm", "achieved": 22.414, "peak": 8000.0, "unit": "GB/s", "frac": 0.002802, "traffic": 41688374897.0,
            "kernel": "xlz::xlz_decode_kernel", "kernel_ms": 270.123, "kernel_ms_median": 270.001, "kernel_ms_steps": [270.1] * 20,
            "algorithmic_bytes_per_launch": 6059123456, "units_per_launch": 65536,
            "traffic_from_profile": {"source": "profiles/r04/%s_pmc.csv" % name, "kernel_rev": "0123456789ab", "fetch": 1.0, "write": 2.0,
                                     "fetch_correction": 1},
            "issue": {"slot_occupancy": 0.9512, "slots": 5120, "lds_bytes_per_slot": 7416, "units": 65536, "rounds": 12.8,
                      "launch_span_ms": 270.0, "unit_ms": {"min": 1.0, "p50": 20.0, "p90": 22.0, "p99": 25.0, "max": 30.0},
                      "corr_queue_key_vs_time": 0.23, "source": "x" * 150, "achieved_decoded_GBps": 15.9, "issue_bound_GBps": 19.8,
                      "issue_bound_from": {"salu_per_decoded_byte": 30.2, "note": "y" * 300}, "latency_bound_GBps": 23.6,
                      "latency_bound_from": {"note": "z" * 300}, "hbm_bound_GBps": 5664.1, "frac_of_bound": 0.8012,
                      "binding": "issue", "bounds_known": ["issue", "latency", "hbm"],
                      "from_profile": {"instructions_per_decoded_byte": 69.7, "budget": "w" * 200}}}


def test_the_bench_line_is_compact_and_round_trips():
    """the LAST stdout line must be one small JSON object with value / config (naming cfg3) / roofline /
    cpu_baseline; everything else lives in the sidecar"""
    import json
    cpu = {"value": 1.0213, "unit": "GiB/s", "cores": 16, "kind": "port", "per_core_mib_s": 65.4, "go_toolchain_on_this_box": False,
           "sample_short": "s" * 180, "sample": "t" * 500}
    spec = bench.CONFIGS["cfg3"]
    full = {"metric": "decompressed GiB/s (aggregate batch)", "value": 14.7654, "unit": "GiB/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 270.913, "ms_per_step_median": 270.8, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": bench.workload_text("cfg3", spec, per_gpu=False) + "; ONE seeded batch split by stream over 1 GPU",
                       "streams_total": 65536, "streams_largest_shard": 65536, "bytes_per_stream": 65536, "compression_ratio": 0.4098,
                       "bit_exact": "all", "parallelism": "one batch sharded by stream x1 (partition_by_weight), no collective",
                       "corpus_generation_s": 100.0, "kernel_rev": "0123456789ab", "library": {"path": "p" * 100, "sha256": "0" * 64}},
            "roofline": _stub_roofline("cfg3"), "cpu_baseline": cpu, "cpu_sanity_liblzma": dict(cpu),
            "timing": {"total_s": 180.0},
            "host_to_host": [{"name": n, "value": 12.5, "phases_ms": {"note": "n" * 300}} for n in ("cfg3", "cfg2-T", "cfg2-T x4")],
            "stream_count_sweep": {"points": [{"streams": k, "cpu_baseline": dict(cpu)} for k in (64, 256, 1024, 4096)],
                                   "break_even_streams": 256, "note": "n" * 200},
            "containers": [{"name": "xz-blocks", "value": 4.31, "workload": "w" * 300}],
            "scaling_projection": {"label": "l" * 100, "n1_value": 17.6, "note": "n" * 300,
                                   "points": [{"n_gpus": k, "streams_on_rank_0": 65536 // k, "wave_slots": 6144, "rounds": 5.33, "ms_per_step": 110.9,
                                               "ms_per_step_wall_mean": 111.2, "steps_ms": [110.9] * 5, "value_projected": 36.057,
                                               "efficiency_projected": 1.0242} for k in (2, 4, 8)]},
            "configs": [{"name": n, "workload": bench.workload_text(n, bench.CONFIGS[n]), "baseline_config": bench.CONFIGS[n]["baseline"],
                         "value": 17.2412, "value_wall": 17.1234, "kernel_ms": 232.123, "roofline": _stub_roofline(n), "cpu_baseline": dict(cpu)}
                        for n in bench.SIDE_ALL]}
    assert len(json.dumps(full)) > 20000          # the detail record is the big one ...
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096, len(text)            # ... the printed line is not
    assert "\n" not in text
    back = json.loads(text)
    assert back == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["config"]["workload"].startswith("cfg3") and "65536" in back["config"]["workload"]
    assert "model" not in back["config"] and back["config"]["kernel_rev"] == "0123456789ab"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
              "units_per_launch", "frac_of_bound", "binding"):
        assert k in back["roofline"], k
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert [c["name"] for c in back["configs"]] == bench.SIDE_ALL
    assert [pt[0] for pt in back["scaling_projection"]["points"]] == [2, 4, 8] and "one GPU" in back["scaling_projection"]["label"]
    # (value = bytes / HIP-event time, value_wall = bytes / host wall time: ADVICE r4 -- two definitions, two keys)
    assert all(set(c) == {"name", "value", "value_wall", "kernel_ms", "frac", "frac_of_bound", "binding", "cpu", "cpu_cores"} for c in back["configs"])
    assert "value_wall" in back["configs_unit"]
    # an N > 1 line (no side configs, no cpu baseline) is a valid line too
    multi = dict(full, n_gpus=8, cpu_baseline=None, configs=None, host_to_host=None, stream_count_sweep=None, containers=None,
                 scaling_projection=None)
    small = bench.compact_line(multi)
    assert small["cpu_baseline"] is None and "configs" not in small and len(json.dumps(small)) < 2500


def test_corpora_can_be_started_early_and_collected_later():
    """bench.py hands every corpus to the pool before the GPU is touched and collects it when its leg begins"""
    with ProcessPoolExecutor(max_workers=2) as pool:
        a = bench.start_corpus(pool, _tiny("cfg2-T", streams=6, size=9000), 1)
        b = bench.start_corpus(pool, _tiny("cfg4", segments=5, size=7000), 1)
        cb, db = b.result()
        ca, da = a.result()
        assert a.result() is a.done   # a second call does not wait again
    assert len(ca) == 6 and len(cb) == 1
    assert hashlib.sha256(lzma.decompress(ca[3], format=lzma.FORMAT_ALONE)).digest() == da[3]
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5 and bench.median([]) is None
