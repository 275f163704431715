"""Builds lzma_amd/libxlz.so (host C ABI + gfx950 kernels) with hipcc, in-tree.

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libxlz.so")
SOURCES = ["xlz_kernel.hip", "xlz_host.hip", "xlz_xz.hip", "xlz_7z.hip"]
HEADERS = ["xlz_format.h", "xlz_check.h", "xlz_fastpath.inc", "xlz_fastpath_pb2.inc", "xlz_fastpath_pb2_br.inc", os.path.join("..", "..", "include", "xlz.h")]
ARCH = "gfx950"


KERNEL_FILES = ["xlz_kernel.hip", "xlz_fastpath.inc", "xlz_fastpath_pb2.inc", "xlz_fastpath_pb2_br.inc", "xlz_format.h"]  # what the device code is made of


def source_id(files=None):
    """12 hex digits over every source the library is compiled from (files=KERNEL_FILES: over the device code only):
    compiled into the library (xlz_build_id / xlz_kernel_id) so that a bench line or a test can tell WHICH kernel ran,
    whatever the file's time stamp says; profiles are tied to the kernel id"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(files or (SOURCES + HEADERS)):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source for gfx950 into lzma_amd/libxlz.so."""
    if out is None and os.environ.get("XLZ_SO"):
        print("lzma_amd.build: XLZ_SO is set, not building: %s" % os.environ["XLZ_SO"], file=sys.stderr)
        return os.environ["XLZ_SO"]  # an A/B build made by hand
    if out is None and not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(HERE, "..", "include"), '-DXLZ_BUILD_ID="%s"' % source_id(), '-DXLZ_KERNEL_ID="%s"' % source_id(KERNEL_FILES)]
    cmd += list(extra_flags)
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    cmd += ["-o", out or SO]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out or SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
