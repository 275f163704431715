"""ctypes binding of libxlz.so (the C ABI declared in include/xlz.h).

There is no Python or CPU decode path behind this module: if the HIP library is
missing or no GPU is usable, calls fail loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(_HERE, "libxlz.so")
SO_PATH = os.environ.get("XLZ_SO") or DEFAULT_SO  # XLZ_SO: A/B builds (announced on stderr when used; bench.py refuses it)

# status codes (include/xlz.h)
OK = 0
OK_INPUT_EOF = 1
ERR_RESULT = -1
ERR_PROPS = -2
ERR_HEADER_EOF = -3
ERR_RC_INIT = -4
ERR_UNEXPECTED_EOF = -5
ERR_OUT_CAP = -6
ERR_BAD_ARG = -7
ERR_DEVICE = -8
ERR_UNSUPPORTED = -9
ERR_CLOSED = -10
ERR_NEED_ONE_READER = -11
ERR_INSUFFICIENT_PROPS = -12
EOF = 100
NEED_INPUT = 101

FMT_LZMA_ALONE = 0
FMT_LZMA_RAW = 1
FMT_LZMA2_RAW = 2

UNKNOWN_SIZE = 0xFFFFFFFFFFFFFFFF

# every symbol include/xlz.h declares (tests check the .so exports all of them)
EXPORTS = [
    "xlz_version", "xlz_build_id", "xlz_kernel_id", "xlz_strerror", "xlz_device_count", "xlz_decode_prop", "xlz_decode_dict_size",
    "xlz_decode_dict_size2", "xlz_decode_unpack_size", "xlz_ctx_create", "xlz_ctx_destroy",
    "xlz_ctx_device", "xlz_ctx_event_record", "xlz_ctx_event_elapsed_ms", "xlz_ctx_enable_batching",
    "xlz_ctx_batching_stats", "xlz_decode_batch", "xlz_ctx_last_call_stats", "xlz_batch_create", "xlz_batch_run", "xlz_batch_sync",
    "xlz_batch_results", "xlz_batch_download", "xlz_batch_device_output", "xlz_batch_last_kernel_ms",
    "xlz_batch_stats", "xlz_batch_destroy", "xlz_new_reader1", "xlz_new_reader2",
    "xlz_new_lzma_decompressor_for_sevenzip", "xlz_new_lzma2_decompressor_for_sevenzip",
    "xlz_reader_read", "xlz_reader_close", "xlz_reader_free", "xlz_xz_index", "xlz_xz_decode",
    "xlz_decode_batch_multi", "xlz_decode_batch_multi_plan", "xlz_xz_decode_multi", "xlz_7z_decode_multi", "xlz_batch_unit_trace", "xlz_reader_stats", "xlz_reader_memory", "xlz_batch_advice", "xlz_batch_launch_info", "xlz_reader_reset", "xlz_reader_reopen", "xlz_reader_expect_more", "xlz_reader_feed", "xlz_reader_feed_eof", "xlz_7z_index", "xlz_7z_decode",
    "xlz_lzma2_units", "xlz_ctx_set_slicing", "xlz_ctx_trim", "xlz_batch_kernel_name", "xlz_decode_batch_plan",
]


class StreamDesc(ctypes.Structure):
    _fields_ = [
        ("inp", ctypes.c_void_p),
        ("in_len", ctypes.c_size_t),
        ("out", ctypes.c_void_p),
        ("out_cap", ctypes.c_size_t),
        ("format", ctypes.c_uint32),
        ("dict_size", ctypes.c_uint32),
        ("unpack_size", ctypes.c_uint64),
        ("props", ctypes.c_uint8),
        ("flags", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8 * 6),
    ]


class Result(ctypes.Structure):
    _fields_ = [
        ("out_len", ctypes.c_uint64),
        ("in_consumed", ctypes.c_uint64),
        ("status", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class CallStats(ctypes.Structure):
    _fields_ = [("upload_ms", ctypes.c_double), ("decode_ms", ctypes.c_double), ("download_ms", ctypes.c_double),
                ("total_ms", ctypes.c_double), ("kernel_span_ms", ctypes.c_double), ("slot_occupancy", ctypes.c_double),
                ("streams", ctypes.c_uint64), ("units", ctypes.c_uint64), ("wave_slots", ctypes.c_uint32),
                ("sub_batches", ctypes.c_uint32), ("slices", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class XzBlock(ctypes.Structure):
    _fields_ = [
        ("comp_off", ctypes.c_uint64),
        ("comp_len", ctypes.c_uint64),
        ("uncomp_off", ctypes.c_uint64),
        ("uncomp_len", ctypes.c_uint64),
        ("check_off", ctypes.c_uint64),
        ("dict_size", ctypes.c_uint32),
        ("check_type", ctypes.c_uint32),
    ]


class Lzma2Unit(ctypes.Structure):
    _fields_ = [
        ("in_off", ctypes.c_uint64),
        ("in_len", ctypes.c_uint64),
        ("out_off", ctypes.c_uint64),
        ("out_len", ctypes.c_uint64),
        ("have_reader", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]


class MultiItem(ctypes.Structure):
    _fields_ = [
        ("stream", ctypes.c_uint64),
        ("in_off", ctypes.c_uint64),
        ("in_len", ctypes.c_uint64),
        ("out_off", ctypes.c_uint64),
        ("out_len", ctypes.c_uint64),
        ("context", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
    ]


class Advice(ctypes.Structure):
    _fields_ = [
        ("units", ctypes.c_uint64),
        ("in_bytes", ctypes.c_uint64),
        ("wave_slots", ctypes.c_uint32),
        ("break_even_units", ctypes.c_uint32),
        ("fill", ctypes.c_double),
        ("prefer_cpu", ctypes.c_int32),
        ("reserved", ctypes.c_uint32),
        ("cpu_cost", ctypes.c_double),
        ("gpu_cost", ctypes.c_double),
    ]


class SzFolder(ctypes.Structure):
    _fields_ = [
        ("pack_off", ctypes.c_uint64),
        ("pack_len", ctypes.c_uint64),
        ("unpack_off", ctypes.c_uint64),
        ("unpack_len", ctypes.c_uint64),
        ("method", ctypes.c_uint32),
        ("dict_size", ctypes.c_uint32),
        ("crc", ctypes.c_uint32),
        ("first_substream", ctypes.c_uint32),
        ("n_substreams", ctypes.c_uint32),
        ("props", ctypes.c_uint8),
        ("has_crc", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8 * 2),
    ]


class SzSubstream(ctypes.Structure):
    _fields_ = [("size", ctypes.c_uint64), ("crc", ctypes.c_uint32), ("has_crc", ctypes.c_uint32)]


_lib = None


def lib():
    """Load libxlz.so; raises if it has not been built (lzma_amd.build.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "lzma_amd: %s is missing -- build the HIP extension first "
            "(python -m lzma_amd.build); there is no CPU fallback" % SO_PATH)
    if SO_PATH != DEFAULT_SO:
        import sys
        print("lzma_amd: XLZ_SO is set -- loading %s instead of the in-tree library" % SO_PATH, file=sys.stderr)
    L = ctypes.CDLL(SO_PATH)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.xlz_version.restype = ctypes.c_char_p
    L.xlz_build_id.restype = ctypes.c_char_p
    L.xlz_kernel_id.restype = ctypes.c_char_p
    L.xlz_strerror.restype = ctypes.c_char_p
    L.xlz_strerror.argtypes = [i32]
    L.xlz_device_count.restype = i32
    L.xlz_decode_prop.argtypes = [ctypes.c_uint8] + [ctypes.POINTER(ctypes.c_uint8)] * 3
    L.xlz_decode_dict_size.restype = ctypes.c_uint32
    L.xlz_decode_dict_size.argtypes = [ctypes.c_char_p]
    L.xlz_decode_dict_size2.restype = ctypes.c_uint32
    L.xlz_decode_dict_size2.argtypes = [ctypes.c_uint8]
    L.xlz_decode_unpack_size.restype = ctypes.c_uint64
    L.xlz_decode_unpack_size.argtypes = [ctypes.c_char_p]
    L.xlz_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.xlz_ctx_destroy.argtypes = [vp]
    L.xlz_ctx_destroy.restype = None
    L.xlz_ctx_device.argtypes = [vp]
    L.xlz_ctx_enable_batching.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32]
    L.xlz_ctx_batching_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.xlz_ctx_event_record.argtypes = [vp, i32]
    L.xlz_ctx_event_elapsed_ms.argtypes = [vp, i32, i32, ctypes.POINTER(ctypes.c_float)]
    L.xlz_decode_batch.argtypes = [vp, ctypes.POINTER(StreamDesc), sz, ctypes.POINTER(Result)]
    L.xlz_ctx_last_call_stats.argtypes = [vp, ctypes.POINTER(CallStats)]
    L.xlz_ctx_set_slicing.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
    L.xlz_ctx_trim.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.xlz_decode_batch_plan.argtypes = [ctypes.POINTER(StreamDesc), sz, ctypes.POINTER(sz), sz, ctypes.POINTER(sz), ctypes.POINTER(i32)]
    L.xlz_batch_kernel_name.argtypes = [vp]
    L.xlz_batch_kernel_name.restype = ctypes.c_char_p
    L.xlz_batch_create.argtypes = [vp, ctypes.POINTER(StreamDesc), sz, ctypes.POINTER(vp)]
    L.xlz_batch_run.argtypes = [vp]
    L.xlz_batch_sync.argtypes = [vp]
    L.xlz_batch_results.argtypes = [vp, ctypes.POINTER(Result)]
    L.xlz_batch_download.argtypes = [vp, sz, vp, sz]
    L.xlz_batch_device_output.argtypes = [vp, sz, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.xlz_batch_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.xlz_batch_stats.argtypes = [vp] + [ctypes.POINTER(ctypes.c_uint64)] * 3
    L.xlz_batch_destroy.argtypes = [vp]
    L.xlz_batch_destroy.restype = None
    L.xlz_new_reader1.restype = vp
    L.xlz_new_reader1.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(i32)]
    L.xlz_new_reader2.restype = vp
    L.xlz_new_reader2.argtypes = [vp, ctypes.c_char_p, sz, i32, ctypes.POINTER(i32)]
    for f in (L.xlz_new_lzma_decompressor_for_sevenzip, L.xlz_new_lzma2_decompressor_for_sevenzip):
        f.restype = vp
        f.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_uint64, ctypes.POINTER(ctypes.c_char_p),
                      ctypes.POINTER(sz), sz, ctypes.POINTER(i32)]
    L.xlz_reader_read.restype = ctypes.c_long
    L.xlz_reader_read.argtypes = [vp, vp, sz, ctypes.POINTER(i32)]
    L.xlz_reader_close.argtypes = [vp]
    L.xlz_reader_free.argtypes = [vp]
    L.xlz_reader_free.restype = None
    L.xlz_reader_reset.argtypes = [vp]
    L.xlz_reader_reopen.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_uint64]
    L.xlz_reader_expect_more.argtypes = [vp]
    L.xlz_reader_feed.argtypes = [vp, ctypes.c_char_p, sz]
    L.xlz_reader_feed_eof.argtypes = [vp]
    L.xlz_reader_stats.argtypes = [vp] + [ctypes.POINTER(ctypes.c_uint64)] * 3
    L.xlz_reader_memory.argtypes = [vp] + [ctypes.POINTER(ctypes.c_uint64)] * 2
    L.xlz_batch_advice.argtypes = [vp, ctypes.POINTER(StreamDesc), sz, ctypes.c_uint32, ctypes.POINTER(Advice)]
    L.xlz_batch_launch_info.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.xlz_batch_unit_trace.argtypes = [vp, vp, vp, vp, sz, ctypes.POINTER(sz)]
    L.xlz_decode_batch_multi.argtypes = [ctypes.POINTER(vp), sz, ctypes.POINTER(StreamDesc), sz, ctypes.POINTER(Result)]
    L.xlz_decode_batch_multi_plan.argtypes = [sz, ctypes.POINTER(StreamDesc), sz, ctypes.POINTER(MultiItem), sz, ctypes.POINTER(sz)]
    L.xlz_xz_index.argtypes = [vp, sz, ctypes.POINTER(XzBlock), sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint64)]
    L.xlz_lzma2_units.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(Lzma2Unit), sz, ctypes.POINTER(sz)]
    L.xlz_7z_index.argtypes = [vp, vp, sz, ctypes.POINTER(SzFolder), sz, ctypes.POINTER(sz), ctypes.POINTER(SzSubstream), sz,
                               ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint64)]
    L.xlz_7z_decode.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint64), i32, ctypes.POINTER(sz)]
    L.xlz_xz_decode.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint64), i32, ctypes.POINTER(sz)]
    L.xlz_xz_decode_multi.argtypes = [ctypes.POINTER(vp), sz, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint64), i32, ctypes.POINTER(sz)]
    L.xlz_7z_decode_multi.argtypes = [ctypes.POINTER(vp), sz, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint64), i32, ctypes.POINTER(sz)]
    _lib = L
    return L


def strerror(status):
    return lib().xlz_strerror(status).decode()


def library_info():
    """which binary is loaded: path, SHA-256 of the file, the source hash compiled into it and the hash of the
    sources in the tree now (equal unless the library is stale or was swapped with XLZ_SO)"""
    import hashlib
    from . import build
    with open(SO_PATH, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    bid = lib().xlz_build_id().decode()
    tree = build.source_id()
    return {"path": os.path.relpath(SO_PATH, os.path.dirname(_HERE)) if SO_PATH.startswith(os.path.dirname(_HERE)) else SO_PATH,
            "sha256": sha, "build_id": bid, "kernel_id": lib().xlz_kernel_id().decode(), "tree_source_id": tree,
            "built_from_tree": bid == tree,
            "xlz_so_override": SO_PATH != DEFAULT_SO}
