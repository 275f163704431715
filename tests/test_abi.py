"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/xlz.h declares.  No compute calls here (there is no GPU in this container)."""
import ctypes
import os
import re

import lzma_amd
from lzma_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "xlz.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xlz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(xlz_so):
    L = ctypes.CDLL(xlz_so)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "libxlz.so does not export " + name
    assert sorted(N.EXPORTS) == declared, "lzma_amd._native.EXPORTS out of date with include/xlz.h"


def test_status_codes_agree_with_oracle_and_header(xlz_so):
    import oracle
    text = open(os.path.join(ROOT, "include", "xlz.h")).read()
    hdr = {m.group(1): int(m.group(2)) for m in re.finditer(r"XLZ_(\w+)\s*=\s*(-?\d+)", text)}
    for name in ["OK", "OK_INPUT_EOF", "ERR_RESULT", "ERR_PROPS", "ERR_HEADER_EOF", "ERR_RC_INIT",
                 "ERR_UNEXPECTED_EOF", "ERR_OUT_CAP", "ERR_BAD_ARG"]:
        assert hdr[name] == getattr(oracle, name) == getattr(N, name), name


def test_helpers_match_reference_semantics(xlz_so):
    # DecodeProp (reader1.go:210-221): 0x5d -> lc3 lp0 pb2; 0x37 -> lc1 lp1 pb1 (the asset whose
    # NAME says lc2)
    assert lzma_amd.DecodeProp(0x5D) == (3, 2, 0)
    assert lzma_amd.DecodeProp(0x37) == (1, 1, 1)
    assert lzma_amd.DecodeProp(0x38) == (2, 1, 1)
    try:
        lzma_amd.DecodeProp(225)
        assert False
    except lzma_amd.LzmaError as e:
        assert e.status == lzma_amd.ERR_PROPS
    assert lzma_amd.DecodeDictSize(b"\x00\x00\x01\x00") == 65536
    assert lzma_amd.DecodeDictSize(b"\x01\x00\x00\x00") == 4096          # clamp, reader1.go:199-201
    assert lzma_amd.DecodeDictSize2(0) == 4096                            # reader2.go:296-298
    assert lzma_amd.DecodeDictSize2(1) == 6144
    assert lzma_amd.DecodeDictSize2(24) == 16 << 20
    assert lzma_amd.DecodeUnpackSize(b"\x47\x01\0\0\0\0\0\0") == 327
    assert lzma_amd.DecodeUnpackSize(b"\xff" * 8) == lzma_amd.UNKNOWN_SIZE
    assert N.strerror(lzma_amd.ERR_RESULT) == "result error"              # errors.go:8
    assert N.lib().xlz_version().startswith(b"xlz")


def test_no_device_fails_loudly(xlz_so):
    """Without a GPU the product path must fail, not fall back to a CPU decoder."""
    if N.lib().xlz_device_count() > 0:
        return
    try:
        lzma_amd.Context(0)
        assert False, "Context() must raise without a HIP device"
    except lzma_amd.LzmaError as e:
        assert e.status == lzma_amd.ERR_DEVICE


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lzma_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "xlz_oracle" not in text and "import oracle" not in text, f


def test_xlz_so_override_loads_another_build(xlz_so, tmp_path):
    """XLZ_SO (README: development aid for A/B measurements) makes the binding load the library it
    names; checked in a child process with a copy of the built library."""
    import os
    import shutil
    import subprocess
    import sys
    other = str(tmp_path / "libxlz_other.so")
    shutil.copy(xlz_so, other)
    code = "import lzma_amd._native as N; N.lib(); print(N.SO_PATH); print(N.lib().xlz_version().decode())"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(os.environ, XLZ_SO=other))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] == other and "xlz" in r.stdout


def test_library_was_built_from_the_sources_in_the_tree(xlz_so):
    """xlz_build_id() is a hash of every source file, compiled into the binary (lzma_amd/build.py): a stale or swapped
    libxlz.so cannot pass for the kernel in the tree -- bench.py quotes the same id and the file's SHA-256"""
    info = N.library_info()
    assert info["build_id"] == info["tree_source_id"] and info["built_from_tree"], info
    assert len(info["sha256"]) == 64 and not info["xlz_so_override"]
