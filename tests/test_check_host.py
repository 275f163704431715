"""Host integrity checks of the container front-ends (lzma_amd/csrc/xlz_check.h: CRC32, CRC64-XZ, SHA-256): known
answers and a bit-by-bit restatement on many lengths and alignments, with and without the carry-less-multiply
folding.  CPU only (g++); the reference has no container code -- what pins these is the published check values."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_crc_and_sha_self_test(tmp_path):
    exe = str(tmp_path / "check_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "lzma_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "check_selftest.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok")
