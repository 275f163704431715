"""A plain C program (gcc, C99, only include/xlz.h) linked against libxlz.so: the stand-in for the
cgo caller of INTEGRATION.md that this image allows (no Go toolchain).  CPU: it builds, links and
fails loudly without a GPU.  GPU: NewReader1 + io.Copy over the reference's a.lzma."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo(xlz_so, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("c") / "reader_demo")
    so_dir = os.path.dirname(xlz_so)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "reader_demo.c"), "-o", exe, "-L", so_dir, "-l:libxlz.so",
                           "-Wl,-rpath," + so_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _fnv(data):
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_program_links_and_fails_loudly_without_a_gpu(demo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([demo, os.path.join(ROOT, "tests", "golden", "a.lzma")], capture_output=True, text=True)
    assert r.returncode == 3 and "HIP device error" in r.stdout


@pytest.mark.gpu
def test_c_program_reads_the_reference_assets(demo, golden):
    exp, data = golden
    for name, bufsz, piece in (("a.lzma", 32768, 262144), ("a_eos.lzma", 7, 64), ("a_lp1_lc2_pb1.lzma", 1, 262144),
                               ("randomfile.dat.lzma", 4096, 100_000)):   # (a 1 MiB file pulled in 100 kB pieces)
        r = subprocess.run([demo, os.path.join(ROOT, "tests", "golden", name), str(bufsz), str(piece)], capture_output=True,
                           text=True)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        word, total, h = r.stdout.split()
        assert word == "EOF" and int(total) == exp[name]["out_len"]
        import oracle
        assert int(h, 16) == _fnv(oracle.lzma1_alone(data[name], 2 << 20)[0])
    r = subprocess.run([demo, os.path.join(ROOT, "tests", "golden", "bad_corrupted.lzma")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("result error")   # reader1_test.go:50-55: Read fails
