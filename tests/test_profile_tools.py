"""CPU: the bookkeeping behind profiles/current.json (tools/save_profile.py, tools/save_profiles_all.py) -- how a
rocprofv3 run over ALL bench.py configs is cut into per-config records, and what the registered entry is made of."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import save_profile as SP
import save_profiles_all as SPA


def test_the_command_line_names_the_legs_and_their_launch_counts():
    legs = SPA.parse_command("bench args: --headline cfg3 --configs cfg2-T,cfg4-R,cfg3 --extras none --steps 3 --warmup 1 "
                             "--side-steps 2 --no-cpu-baseline")
    assert legs == [("cfg3", 4), ("cfg2-T", 3), ("cfg4-R", 3)]   # the headline is never a side config as well
    cut = SPA.split(list(range(10)), legs, "test")
    assert cut == {"cfg3": [0, 1, 2, 3], "cfg2-T": [4, 5, 6], "cfg4-R": [7, 8, 9]}
    with pytest.raises(AssertionError):   # one launch more than the command line makes: nothing may be guessed
        SPA.split(list(range(11)), legs, "test")


def test_an_entry_prices_traffic_and_issue_from_the_counters():
    vals = {"FETCH_SIZE": [1000.0, 1000.0], "WRITE_SIZE": [500.0], "SQ_INSTS_SALU": [2.0e9], "SQ_INSTS_VALU": [1.0e9],
            "SQ_INSTS_BRANCH": [1e8], "SQ_INSTS_LDS": [1e8], "SQ_INSTS_VMEM": [1e7], "SQ_WAVE_CYCLES": [4096 * 2.4e6 / 4],
            "SQ_ACTIVE_INST_ANY": [1.0], "SQ_WAIT_ANY": [1.0], "SQ_WAIT_INST_ANY": [2.0]}
    e = SP.make_entry("w", "abc", decoded=1e8, kernel_ms=1.0, grid_threads=4096 * 64, vals=vals, dest="p/x")
    assert e["fetch_bytes_per_launch_raw"] == 1000 * 1024 and e["fetch_correction"] == 1
    assert e["traffic_bytes_per_launch"] == 1500 * 1024
    assert e["issue"]["salu_per_decoded_byte"] == 20.0 and e["issue"]["valu_per_decoded_byte"] == 10.0
    assert abs(e["issue"]["slot_occupancy"] - 1.0) < 1e-6     # 4096 slots busy for the whole millisecond
    # the stored-chunk copy reads 16 bytes per lane in streaming order: gfx950 tallies those at half their size
    w = SP.make_entry("w", "abc", decoded=1e8, kernel_ms=1.0, grid_threads=4096 * 64, vals=vals, dest="p/x", wide_reads=True)
    assert w["fetch_correction"] == 2 and w["traffic_bytes_per_launch"] == 2500 * 1024
    assert w["fetch_bytes_per_launch_raw"] == e["fetch_bytes_per_launch_raw"]
