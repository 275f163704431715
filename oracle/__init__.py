"""ctypes binding of the CPU oracle (oracle/xlz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  lzma_amd/ never imports this package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libxlz_oracle.so")

OK = 0
OK_INPUT_EOF = 1
ERR_RESULT = -1
ERR_PROPS = -2
ERR_HEADER_EOF = -3
ERR_RC_INIT = -4
ERR_UNEXPECTED_EOF = -5
ERR_OUT_CAP = -6
ERR_BAD_ARG = -7

FLAG_REF_U16_COMPSIZE = 1


class Result(ctypes.Structure):
    _fields_ = [
        ("out_len", ctypes.c_uint64),
        ("in_consumed", ctypes.c_uint64),
        ("status", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class Job(ctypes.Structure):
    _fields_ = [
        ("inp", ctypes.c_void_p),
        ("in_len", ctypes.c_size_t),
        ("out", ctypes.c_void_p),
        ("out_cap", ctypes.c_size_t),
        ("fmt", ctypes.c_uint32),
        ("dict_size", ctypes.c_uint32),
    ]


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "xlz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.c_char_p
        L.xlzo_lzma1_alone.argtypes = [u8p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.POINTER(Result)]
        L.xlzo_lzma1_raw.argtypes = [ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint64, u8p,
                                     ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.POINTER(Result)]
        L.xlzo_lzma2_raw.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u8p, ctypes.c_size_t,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Result)]
        L.xlzo_decode_batch_mt.argtypes = [ctypes.POINTER(Job), ctypes.c_size_t, ctypes.c_int,
                                           ctypes.POINTER(Result)]
        L.xlzo_decode_dict_size2.restype = ctypes.c_uint32
        L.xlzo_decode_dict_size2.argtypes = [ctypes.c_uint8]
        _lib = L
    return _lib


def _run(fn, out_cap, *args):
    buf = ctypes.create_string_buffer(max(out_cap, 1))
    res = Result()
    rc = fn(*args, ctypes.cast(buf, ctypes.c_void_p), out_cap, ctypes.byref(res))
    if rc != 0:
        raise ValueError("oracle bad argument (%d)" % rc)
    return buf.raw[: res.out_len], res.status, res.in_consumed


def lzma1_alone(data, out_cap):
    """NewReader1(bytes) + io.Copy -> (output bytes, status, in_consumed)."""
    return _run(lib().xlzo_lzma1_alone, out_cap, data, len(data))


def lzma1_raw(props, dict_size, unpack_size, data, out_cap):
    """NewLZMADecompressorForSevenZip-style: header fields out of band."""
    return _run(lib().xlzo_lzma1_raw, out_cap, props, dict_size, unpack_size, data, len(data))


def lzma2_raw(data, dict_size, out_cap, flags=0):
    """NewReader2(bytes, dictSize) + io.Copy."""
    return _run(lib().xlzo_lzma2_raw, out_cap, dict_size, flags, data, len(data))


def decode_dict_size2(b):
    return lib().xlzo_decode_dict_size2(b)


def decode_batch_mt(streams, out_caps, nthreads, fmt=0, dict_size=0, timing=False):
    """Decode a list of byte strings with nthreads host threads (one stream per thread).

    Returns (list of output bytes, list of statuses); with timing=True the outputs stay
    ctypes buffers (no copies) and a third value is the wall time of the C call alone, which
    is what bench.py's cpu_baseline leg reports.
    """
    import time
    n = len(streams)
    jobs = (Job * n)()
    res = (Result * n)()
    keep = []
    outs = []
    for i, (s, cap) in enumerate(zip(streams, out_caps)):
        ib = ctypes.create_string_buffer(s, len(s)) if not isinstance(s, ctypes.Array) else s
        ob = ctypes.create_string_buffer(max(cap, 1))
        keep.append(ib)
        outs.append(ob)
        jobs[i].inp = ctypes.cast(ib, ctypes.c_void_p)
        jobs[i].in_len = len(s)
        jobs[i].out = ctypes.cast(ob, ctypes.c_void_p)
        jobs[i].out_cap = cap
        jobs[i].fmt = fmt
        jobs[i].dict_size = dict_size
    t0 = time.perf_counter()
    lib().xlzo_decode_batch_mt(jobs, n, nthreads, res)
    dt = time.perf_counter() - t0
    sts = [res[i].status for i in range(n)]
    if timing:
        return [(outs[i], res[i].out_len) for i in range(n)], sts, dt
    return [outs[i].raw[: res[i].out_len] for i in range(n)], sts
