"""CPU: a few seconds of tools/fuzz_parsers.py -- mutated .xz / .7z files through the host-only index
parsers: a status, or byte ranges inside the file; never a crash (the AddressSanitizer run of the
same tool is described in its docstring)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_parsers", os.path.join(ROOT, "tools", "fuzz_parsers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_mutated_xz_files_parse_or_fail_cleanly(xlz_so):
    n, ok = _tool().fuzz_xz(3.0, 2026)
    assert n > 1000 and 0 < ok < n


def test_mutated_7z_archives_parse_or_fail_cleanly(xlz_so):
    n, ok = _tool().fuzz_7z(3.0, 2026)
    assert n > 1000 and 0 < ok < n


def test_lzma2_unit_plans_tile_whatever_the_headers_say():
    n, units = _tool().fuzz_lzma2_units(3.0, 2026)
    assert n > 200 and units >= n


def test_batch_planners_answer_consistently_whatever_the_descriptors_say(xlz_so):
    """xlz_batch_advice, xlz_decode_batch_plan and xlz_decode_batch_multi_plan on descriptors of every format with mutated
    inputs and odd capacities: no device, nothing decoded; the assertions are the tool's own"""
    calls, dealt = _tool().fuzz_plans(4.0, 2026)
    assert calls > 50 and dealt > calls
