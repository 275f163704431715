import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    exp = json.load(open(os.path.join(GOLDEN, "expected.json")))["files"]
    data = {name: open(os.path.join(GOLDEN, name), "rb").read() for name in exp}
    return exp, data


@pytest.fixture(scope="session")
def xlz_so():
    """Build (if stale) and return the path of the HIP library."""
    from lzma_amd import build
    return build.build()


@pytest.fixture(scope="session")
def ctx(xlz_so):
    """A GPU context; the HIP path must be the one that runs (no fallback)."""
    import lzma_amd
    return lzma_amd.Context(0)
