"""lzma_amd -- MI355X-native batched LZMA / LZMA2 decoder.

Host-side mirror of the reference's Go surface (kulaginds/lzma: NewReader1,
NewReader2, the bodgit/sevenzip decompressor constructors and the exported
Decode* helpers) on top of the C ABI in include/xlz.h.  All decoding happens in
hand-written HIP kernels (lzma_amd/csrc); this package only marshals buffers.
It fails loudly when libxlz.so or a GPU is missing -- there is no CPU fallback.
"""
import ctypes

from . import _native as N
from ._native import (EOF, ERR_BAD_ARG, ERR_CLOSED, ERR_DEVICE, ERR_HEADER_EOF,  # noqa: F401
                      ERR_INSUFFICIENT_PROPS, ERR_NEED_ONE_READER, ERR_OUT_CAP, ERR_PROPS,
                      ERR_RC_INIT, ERR_RESULT, ERR_UNEXPECTED_EOF, ERR_UNSUPPORTED,
                      FMT_LZMA2_RAW, FMT_LZMA_ALONE, FMT_LZMA_RAW, NEED_INPUT, OK, OK_INPUT_EOF, UNKNOWN_SIZE)


class LzmaError(Exception):
    """A reference error value (errors.go:5-12 and friends) carried as a status code."""

    def __init__(self, status, where=""):
        self.status = status
        msg = N.strerror(status)
        super().__init__("%s%s" % (where + ": " if where else "", msg))


# the reference's sentinels, as statuses
ErrResultError = ERR_RESULT                  # errors.go:8
ErrIncorrectProperties = ERR_PROPS           # errors.go:7
ErrUnexpectedEOF = ERR_UNEXPECTED_EOF        # io.ErrUnexpectedEOF (reader2.go:104-127)


class Stream:
    """One compressed stream of a batch (xlz_stream_desc)."""

    def __init__(self, data, fmt=FMT_LZMA_ALONE, out_cap=None, dict_size=0, unpack_size=UNKNOWN_SIZE,
                 props=0):
        self.data = bytes(data)
        self.fmt = fmt
        self.dict_size = dict_size
        self.unpack_size = unpack_size
        self.props = props
        if out_cap is None:
            out_cap = self._guess_cap()
        self.out_cap = out_cap

    def _guess_cap(self):
        if self.fmt == FMT_LZMA_ALONE and len(self.data) >= 13:
            u = int.from_bytes(self.data[5:13], "little")
            if u != UNKNOWN_SIZE:
                return u
        if self.fmt == FMT_LZMA_RAW and self.unpack_size != UNKNOWN_SIZE:
            return self.unpack_size
        raise ValueError("out_cap is required when the stream does not carry its size")


class Context:
    """One HIP device + stream (xlz_ctx).  One per host thread and GPU."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        st = N.lib().xlz_ctx_create(device, ctypes.byref(self._h))
        if st != OK:
            raise LzmaError(st, "xlz_ctx_create(device=%d)" % device)

    def enable_batching(self, window_us=500, max_streams=4096):
        """Readers created on this context from now on are decoded together (xlz_ctx_enable_batching)."""
        st = N.lib().xlz_ctx_enable_batching(self._h, window_us, max_streams)
        if st != OK:
            raise LzmaError(st, "xlz_ctx_enable_batching")

    def batching_stats(self):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        st = N.lib().xlz_ctx_batching_stats(self._h, ctypes.byref(a), ctypes.byref(b))
        if st != OK:
            raise LzmaError(st, "xlz_ctx_batching_stats")
        return a.value, b.value

    def last_call_stats(self):
        """phase times and slot occupancy of the last decode_batch on this context (xlz_ctx_last_call_stats) -> dict"""
        cs = N.CallStats()
        st = N.lib().xlz_ctx_last_call_stats(self._h, ctypes.byref(cs))
        if st != OK:
            raise LzmaError(st, "xlz_ctx_last_call_stats")
        return {k: getattr(cs, k) for k, _ in N.CallStats._fields_}

    def set_slicing(self, min_call_bytes=0, slice_bytes=0, max_slices=0):
        """when a decode_batch of one wave round runs as a sequence of launches whose downloads overlap the decode
        (xlz_ctx_set_slicing; 0 = default, max_slices=1: never)"""
        st = N.lib().xlz_ctx_set_slicing(self._h, min_call_bytes, slice_bytes, max_slices)
        if st != OK:
            raise LzmaError(st, "xlz_ctx_set_slicing")

    def trim(self):
        """release the device / pinned memory decode_batch keeps between calls (xlz_ctx_trim) -> bytes released"""
        n = ctypes.c_uint64()
        st = N.lib().xlz_ctx_trim(self._h, ctypes.byref(n))
        if st != OK:
            raise LzmaError(st, "xlz_ctx_trim")
        return n.value

    def event_record(self, slot):
        st = N.lib().xlz_ctx_event_record(self._h, slot)
        if st != OK:
            raise LzmaError(st, "xlz_ctx_event_record")

    def event_elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        st = N.lib().xlz_ctx_event_elapsed_ms(self._h, a, b, ctypes.byref(ms))
        if st != OK:
            raise LzmaError(st, "xlz_ctx_event_elapsed_ms")
        return ms.value

    def close(self):
        if self._h:
            N.lib().xlz_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _make_descs(streams, with_out=True):
    n = len(streams)
    descs = (N.StreamDesc * n)()
    keep_in, outs = [], []
    for i, s in enumerate(streams):
        ib = ctypes.create_string_buffer(s.data, len(s.data)) if len(s.data) else ctypes.create_string_buffer(1)
        keep_in.append(ib)
        descs[i].inp = ctypes.cast(ib, ctypes.c_void_p)
        descs[i].in_len = len(s.data)
        if with_out:
            ob = ctypes.create_string_buffer(max(int(s.out_cap), 1))
            outs.append(ob)
            descs[i].out = ctypes.cast(ob, ctypes.c_void_p)
        descs[i].out_cap = int(s.out_cap)
        descs[i].format = s.fmt
        descs[i].dict_size = s.dict_size & 0xFFFFFFFF
        descs[i].unpack_size = s.unpack_size
        descs[i].props = s.props
    return descs, keep_in, outs


def decode_batch(ctx, streams):
    """Decode independent streams on the GPU.

    Returns a list of (output bytes, status, in_consumed).  A bad stream never
    fails the batch; the call raises only if it could not run at all.
    """
    streams = list(streams)
    n = len(streams)
    if n == 0:
        return []
    descs, keep, outs = _make_descs(streams)
    res = (N.Result * n)()
    st = N.lib().xlz_decode_batch(ctx._h, descs, n, res)
    if st != OK:
        raise LzmaError(st, "xlz_decode_batch")
    del keep
    return [(outs[i].raw[: res[i].out_len], res[i].status, res[i].in_consumed) for i in range(n)]


def decode_batch_plan(out_caps):
    """xlz_decode_batch_plan (host only): how decode_batch would cut a call of streams with these output capacities into
    pieces -> (list of the pieces' first stream indices + [n], mode: 0 one piece, 1 overlapped pieces, 2 one-round pieces)"""
    n = len(out_caps)
    descs = (N.StreamDesc * max(n, 1))()
    for i, c in enumerate(out_caps):
        descs[i].out_cap = int(c)
    cuts = (ctypes.c_size_t * (n + 2))()
    k, mode = ctypes.c_size_t(), ctypes.c_int32()
    st = N.lib().xlz_decode_batch_plan(descs, n, cuts, n + 2, ctypes.byref(k), ctypes.byref(mode))
    if st != OK:
        raise LzmaError(st, "xlz_decode_batch_plan")
    return list(cuts[: k.value]), mode.value


def batch_advice(streams, host_threads=0, ctx=None):
    """xlz_batch_advice: what a decode of `streams` would launch and whether the host's cores are the faster decoder
    for it -- host only, nothing is uploaded.  -> dict (units, in_bytes, wave_slots, break_even_units, fill, prefer_cpu)"""
    streams = list(streams)
    descs, keep, _ = _make_descs(streams, with_out=False)
    adv = N.Advice()
    st = N.lib().xlz_batch_advice(ctx._h if ctx else None, descs, len(streams), host_threads, ctypes.byref(adv))
    if st != OK:
        raise LzmaError(st, "xlz_batch_advice")
    del keep
    return {f: getattr(adv, f) for f, _ in N.Advice._fields_ if f != "reserved"}


def multi_plan(n_ctx, streams):
    """xlz_decode_batch_multi_plan (host only): what decode_batch_on would hand to which of n_ctx contexts -> list of dicts
    (stream, in_off, in_len, out_off, out_len, context, whole, first, last), the items of a stream adjacent and in order"""
    streams = list(streams)
    descs, keep, _ = _make_descs(streams, with_out=False)
    n = ctypes.c_size_t()
    st = N.lib().xlz_decode_batch_multi_plan(n_ctx, descs, len(streams), None, 0, ctypes.byref(n))
    if st != OK:
        raise LzmaError(st, "xlz_decode_batch_multi_plan")
    items = (N.MultiItem * max(n.value, 1))()
    st = N.lib().xlz_decode_batch_multi_plan(n_ctx, descs, len(streams), items, n.value, ctypes.byref(n))
    if st != OK:
        raise LzmaError(st, "xlz_decode_batch_multi_plan")
    del keep
    return [dict(stream=it.stream, in_off=it.in_off, in_len=it.in_len, out_off=it.out_off, out_len=it.out_len, context=it.context,
                 whole=bool(it.flags & 1), first=bool(it.flags & 2), last=bool(it.flags & 4)) for it in items[: n.value]]


def decode_batch_on(ctxs, streams):
    """decode_batch over several contexts (one per GPU) through xlz_decode_batch_multi: sharded by
    stream inside the library, one host thread per context, results in input order."""
    streams = list(streams)
    n = len(streams)
    if n == 0:
        return []
    descs, keep, outs = _make_descs(streams)
    res = (N.Result * n)()
    hs = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    st = N.lib().xlz_decode_batch_multi(hs, len(ctxs), descs, n, res)
    if st != OK:
        raise LzmaError(st, "xlz_decode_batch_multi")
    del keep
    return [(outs[i].raw[: res[i].out_len], res[i].status, res[i].in_consumed) for i in range(n)]


class Batch:
    """Device-resident batch: upload once, run many times (xlz_batch)."""

    def __init__(self, ctx, streams):
        self.ctx = ctx
        self.streams = list(streams)
        self.n = len(self.streams)
        descs, keep, _ = _make_descs(self.streams, with_out=False)
        self._h = ctypes.c_void_p()
        st = N.lib().xlz_batch_create(ctx._h, descs, self.n, ctypes.byref(self._h))
        if st != OK:
            raise LzmaError(st, "xlz_batch_create")

    def run(self):
        st = N.lib().xlz_batch_run(self._h)
        if st != OK:
            raise LzmaError(st, "xlz_batch_run")

    def sync(self):
        st = N.lib().xlz_batch_sync(self._h)
        if st != OK:
            raise LzmaError(st, "xlz_batch_sync")

    def kernel_ms(self):
        ms = ctypes.c_float()
        st = N.lib().xlz_batch_last_kernel_ms(self._h, ctypes.byref(ms))
        if st != OK:
            raise LzmaError(st, "xlz_batch_last_kernel_ms")
        return ms.value

    def results(self):
        res = (N.Result * max(self.n, 1))()
        st = N.lib().xlz_batch_results(self._h, res)
        if st != OK:
            raise LzmaError(st, "xlz_batch_results")
        return [(res[i].out_len, res[i].status, res[i].in_consumed) for i in range(self.n)]

    def stats(self):
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        st = N.lib().xlz_batch_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        if st != OK:
            raise LzmaError(st, "xlz_batch_stats")
        return a.value, b.value, c.value

    def launch_info(self):
        """(resident single-wave workgroups = wave slots, LDS bytes per workgroup) of the decode launch"""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        st = N.lib().xlz_batch_launch_info(self._h, ctypes.byref(a), ctypes.byref(b))
        if st != OK:
            raise LzmaError(st, "xlz_batch_launch_info")
        return a.value, b.value

    def kernel_name(self):
        """the kernel of the batch's main launch (xlz_batch_kernel_name): the full or the compact (pb <= 2) model layout"""
        return N.lib().xlz_batch_kernel_name(self._h).decode()

    def unit_trace(self):
        """(t_start, t_end, in_len) numpy uint32 arrays, one entry per unit of the last run: ticks of
        the device's 100 MHz clock since the first unit started (xlz_batch_unit_trace)."""
        import numpy as np
        n = ctypes.c_size_t()
        st = N.lib().xlz_batch_unit_trace(self._h, None, None, None, 0, ctypes.byref(n))
        if st != OK:
            raise LzmaError(st, "xlz_batch_unit_trace")
        a, b, c = (np.zeros(max(n.value, 1), dtype=np.uint32) for _ in range(3))
        st = N.lib().xlz_batch_unit_trace(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, n.value, ctypes.byref(n))
        if st != OK:
            raise LzmaError(st, "xlz_batch_unit_trace")
        return a[: n.value], b[: n.value], c[: n.value]

    def download(self, i, length):
        buf = ctypes.create_string_buffer(max(int(length), 1))
        st = N.lib().xlz_batch_download(self._h, i, ctypes.cast(buf, ctypes.c_void_p), int(length))
        if st != OK:
            raise LzmaError(st, "xlz_batch_download")
        return buf.raw[:length]

    def device_output(self, i):
        p, cap = ctypes.c_void_p(), ctypes.c_size_t()
        st = N.lib().xlz_batch_device_output(self._h, i, ctypes.byref(p), ctypes.byref(cap))
        if st != OK:
            raise LzmaError(st, "xlz_batch_device_output")
        return p.value, cap.value

    def close(self):
        if self._h:
            N.lib().xlz_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- exported helpers with the reference's names -------------------------------
def DecodeProp(d):
    """reader1.go:210-221 -> (lc, pb, lp); raises LzmaError(ErrIncorrectProperties)."""
    lc, pb, lp = ctypes.c_uint8(), ctypes.c_uint8(), ctypes.c_uint8()
    st = N.lib().xlz_decode_prop(d, ctypes.byref(lc), ctypes.byref(pb), ctypes.byref(lp))
    if st != OK:
        raise LzmaError(st)
    return lc.value, pb.value, lp.value


def DecodeDictSize(properties):
    """reader1.go:193-208"""
    return N.lib().xlz_decode_dict_size(bytes(properties[:4]))


def DecodeDictSize2(encoded):
    """reader2.go:296-298"""
    return N.lib().xlz_decode_dict_size2(encoded)


def DecodeUnpackSize(header):
    """reader1.go:178-191"""
    return N.lib().xlz_decode_unpack_size(bytes(header[:8]))


# ---- pull-style readers ----------------------------------------------------------
class io_EOF:  # sentinel standing in for Go's io.EOF
    pass


class _Reader:
    def __init__(self, ctx, handle, source=None, piece=1 << 20):
        self._ctx = ctx
        self._h = ctypes.c_void_p(handle)
        self._src = source  # file-like object the rest of the compressed stream is pulled from (streaming input)
        self._piece = piece
        if source is not None:
            st = N.lib().xlz_reader_expect_more(self._h)
            if st != OK:
                raise LzmaError(st, "xlz_reader_expect_more")

    def Read(self, n):
        """Go's Read(p []byte): returns (bytes, err) with err None, io_EOF or LzmaError.  With a
        source (NewReader1 / NewReader2 on a file object) the compressed side is pulled piece by piece
        as the decoder asks for it -- what the Go shim does with its io.Reader."""
        buf = ctypes.create_string_buffer(max(n, 1))
        err = ctypes.c_int()
        got = 0
        while True:
            k = N.lib().xlz_reader_read(self._h, ctypes.cast(ctypes.addressof(buf) + got, ctypes.c_void_p), n - got,
                                        ctypes.byref(err))
            got += k
            if err.value != NEED_INPUT or self._src is None:
                break
            more = self._src.read(self._piece)
            st = N.lib().xlz_reader_feed(self._h, more, len(more)) if more else N.lib().xlz_reader_feed_eof(self._h)
            if st != OK:
                return buf.raw[:got], LzmaError(st, "feeding the reader")
            if got == n and n:
                err.value = OK
                break
        e = None
        if err.value == EOF:
            e = io_EOF
        elif err.value != OK:
            e = LzmaError(err.value, "lzma: error reading" if self._is_closer else "")
        return buf.raw[:got], e

    def read_all(self, chunk=32768):
        """io.Copy(dst, r): returns (bytes, err) where err is None at io.EOF."""
        out = []
        while True:
            b, e = self.Read(chunk)
            out.append(b)
            if e is io_EOF:
                return b"".join(out), None
            if e is not None:
                return b"".join(out), e

    def stats(self):
        """(refill launches, whole-stream fallback decodes, compressed bytes uploaded) -- xlz_reader_stats"""
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        N.lib().xlz_reader_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def memory(self):
        """(bytes of the sliding output window on the device, bytes of the window image -- 0 until the stream's first
        dictionary reset behind a non-empty epoch) -- xlz_reader_memory"""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        N.lib().xlz_reader_memory(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def Close(self):
        """readCloser.Close (readcloser.go:16-28)."""
        st = N.lib().xlz_reader_close(self._h)
        if st != OK:
            return LzmaError(st)
        return None

    _is_closer = False

    def __del__(self):
        try:
            if self._h:
                N.lib().xlz_reader_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


class Reader1(_Reader):
    def Reset(self):
        """(*Reader1).Reset (reader1.go:161-164)"""
        st = N.lib().xlz_reader_reset(self._h)
        if st != OK:
            raise LzmaError(st, "Reset")

    def Reopen(self, data, unpack_size=UNKNOWN_SIZE, piece=1 << 20):
        """(*Reader1).Reopen(inStream, unpackSize) (reader1.go:166-176): returns err.  `data`: the new raw stream as
        bytes, or a file-like object that is then pulled `piece` bytes at a time (the reference takes an io.ByteReader)"""
        head, src = _head_and_source(data, piece)
        st = N.lib().xlz_reader_reopen(self._h, head, len(head), unpack_size)
        self._src, self._piece = None, piece
        if st == ERR_HEADER_EOF:
            return io_EOF
        if st != OK:
            return LzmaError(st)
        if src is not None:
            st = N.lib().xlz_reader_expect_more(self._h)
            if st != OK:
                return LzmaError(st, "xlz_reader_expect_more")
            self._src = src
        return None


class Reader2(_Reader):
    pass


class ReadCloser(_Reader):
    _is_closer = True


def _head_and_source(data, piece):
    """bytes -> (bytes, None); a file-like object -> (its first piece, the object or None at its end)"""
    if hasattr(data, "read"):
        head = data.read(piece)
        return head, (data if len(head) == piece else None)
    return bytes(data), None


def NewReader1(ctx, data, piece=1 << 20):
    """NewReader1(inStream) (reader1.go:18-24): returns (reader, err).  `data`: the compressed bytes, or a
    file-like object that is then read `piece` bytes at a time as the decoder needs them."""
    head, src = _head_and_source(data, piece)
    err = ctypes.c_int()
    h = N.lib().xlz_new_reader1(ctx._h, head, len(head), ctypes.byref(err))
    if not h:
        return None, LzmaError(err.value)
    return Reader1(ctx, h, src, piece), None


def NewReader2(ctx, data, dict_size, piece=1 << 20):
    """NewReader2(inStream, dictSize) (reader2.go:26-41): returns (reader, err); `data` as for NewReader1."""
    head, src = _head_and_source(data, piece)
    err = ctypes.c_int()
    h = N.lib().xlz_new_reader2(ctx._h, head, len(head), dict_size, ctypes.byref(err))
    if not h:
        return None, LzmaError(err.value)
    return Reader2(ctx, h, src, piece), None


def _sevenzip(fn, ctx, props, unpack_size, readers):
    n = len(readers)
    arr = (ctypes.c_char_p * max(n, 1))(*[bytes(r) for r in readers])
    lens = (ctypes.c_size_t * max(n, 1))(*[len(r) for r in readers])
    err = ctypes.c_int()
    h = fn(ctx._h, bytes(props), len(props), unpack_size, arr, lens, n, ctypes.byref(err))
    if not h:
        return None, LzmaError(err.value)
    return ReadCloser(ctx, h), None


def NewLZMADecompressorForSevenZip(ctx, props, unpack_size, readers):
    """reader1.go:32-61: props = props byte + LE32 dict size; exactly one reader."""
    return _sevenzip(N.lib().xlz_new_lzma_decompressor_for_sevenzip, ctx, props, unpack_size, readers)


def NewLZMA2DecompressorForSevenZip(ctx, props, unpack_size, readers):
    """reader2.go:49-75: props = one dict-size byte; exactly one reader."""
    return _sevenzip(N.lib().xlz_new_lzma2_decompressor_for_sevenzip, ctx, props, unpack_size, readers)


def lzma2_units(data):
    """The unit plan of a raw LZMA2 stream (xlz_lzma2_units; host only): list of dicts (in_off, in_len, out_off,
    out_len, have_reader) -- what a decode of `data` as FMT_LZMA2_RAW launches, one wave per unit."""
    if hasattr(data, "ctypes"):   # a numpy array (uint8, contiguous): planned in place, whatever its size
        ptr, size = ctypes.c_char_p(data.ctypes.data), data.size
    else:
        data = bytes(data)
        ptr, size = data, len(data)
    n = ctypes.c_size_t()
    st = N.lib().xlz_lzma2_units(ptr, size, None, 0, ctypes.byref(n))
    if st != OK:
        raise LzmaError(st, "xlz_lzma2_units")
    units = (N.Lzma2Unit * max(n.value, 1))()
    st = N.lib().xlz_lzma2_units(ptr, size, units, n.value, ctypes.byref(n))
    if st != OK:
        raise LzmaError(st, "xlz_lzma2_units")
    return [{f: getattr(units[k], f) for f, _ in N.Lzma2Unit._fields_ if f != "reserved"} for k in range(n.value)]


# ---- .xz container front-end (include/xlz.h: xlz_xz_index / xlz_xz_decode) -------------------
def xz_index(data):
    """Block index of an .xz file: list of dicts (comp_off, comp_len, uncomp_off, uncomp_len,
    dict_size, check_type) and the total decoded size.  Host only."""
    buf = ctypes.create_string_buffer(data, len(data)) if len(data) else ctypes.create_string_buffer(1)
    n = ctypes.c_size_t()
    total = ctypes.c_uint64()
    st = N.lib().xlz_xz_index(ctypes.cast(buf, ctypes.c_void_p), len(data), None, 0, ctypes.byref(n), ctypes.byref(total))
    if st != OK:
        raise LzmaError(st, "xlz_xz_index")
    blocks = (N.XzBlock * max(n.value, 1))()
    st = N.lib().xlz_xz_index(ctypes.cast(buf, ctypes.c_void_p), len(data), blocks, n.value, ctypes.byref(n),
                              ctypes.byref(total))
    if st != OK:
        raise LzmaError(st, "xlz_xz_index")
    fields = [f for f, _ in N.XzBlock._fields_]
    return [{f: getattr(blocks[i], f) for f in fields} for i in range(n.value)], total.value


def xz_decode(ctx, data, verify=True, max_size=None):
    """Decode a whole .xz file (all streams, all blocks) as one GPU batch -> bytes.
    verify: check every block's CRC32 / CRC64.  max_size: refuse (ERR_OUT_CAP) a file whose index announces more."""
    _, total = xz_index(data)
    if max_size is not None and total > max_size:
        raise LzmaError(ERR_OUT_CAP, "xlz_xz_decode: the index announces %d bytes, max_size is %d" % (total, max_size))
    out = ctypes.create_string_buffer(max(total, 1))
    n = xz_decode_into(ctx, data, out, verify=verify)
    return out.raw[:n]


def xz_decode_on(ctxs, data, verify=True):
    """xlz_xz_decode_multi: the file's blocks -- and the units inside large blocks -- dealt to several contexts -> bytes"""
    _, total = xz_index(data)
    out = ctypes.create_string_buffer(max(total, 1))
    hs = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    out_len, unverified = ctypes.c_uint64(), ctypes.c_size_t()
    data = bytes(data)
    st = N.lib().xlz_xz_decode_multi(hs, len(ctxs), ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p), len(data),
                                     ctypes.cast(out, ctypes.c_void_p), total, ctypes.byref(out_len), 1 if verify else 0,
                                     ctypes.byref(unverified))
    if st != OK:
        raise LzmaError(st, "xlz_xz_decode_multi")
    return out.raw[: out_len.value]


def xz_decode_into(ctx, data, out, verify=True):
    """xlz_xz_decode with the caller's buffers and nothing else: `data` is read in place (bytes, or any object with the
    buffer interface), the decoded bytes land in `out` (a writable buffer of at least the index's total: bytearray,
    numpy array, ctypes array) -> number of bytes decoded.  What bench.py times for the container line."""
    if not isinstance(data, bytes):
        data = bytes(data)
    src_ptr = ctypes.c_char_p(data)  # the library only reads it: no copy
    dst = out if isinstance(out, ctypes.Array) else (ctypes.c_char * memoryview(out).nbytes).from_buffer(out)
    out_len = ctypes.c_uint64()
    unverified = ctypes.c_size_t()
    st = N.lib().xlz_xz_decode(ctx._h, ctypes.cast(src_ptr, ctypes.c_void_p), len(data), ctypes.cast(dst, ctypes.c_void_p),
                               ctypes.sizeof(dst), ctypes.byref(out_len), 1 if verify else 0, ctypes.byref(unverified))
    if st != OK:
        raise LzmaError(st, "xlz_xz_decode")
    return out_len.value


def sevenzip_decode_on(ctxs, data, verify=True):
    """xlz_7z_decode_multi: the archive's folders dealt to several contexts -> the files' bytes back to back"""
    _, _, total = sevenzip_index(data, ctxs[0])
    buf = ctypes.create_string_buffer(data, len(data))
    out = ctypes.create_string_buffer(max(total, 1))
    hs = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    out_len, unverified = ctypes.c_uint64(), ctypes.c_size_t()
    st = N.lib().xlz_7z_decode_multi(hs, len(ctxs), ctypes.cast(buf, ctypes.c_void_p), len(data), ctypes.cast(out, ctypes.c_void_p),
                                     total, ctypes.byref(out_len), 1 if verify else 0, ctypes.byref(unverified))
    if st != OK:
        raise LzmaError(st, "xlz_7z_decode_multi")
    return out.raw[: out_len.value]


# ---- .7z container front-end (include/xlz.h: xlz_7z_index / xlz_7z_decode) -------------------
def sevenzip_index(data, ctx=None):
    """Folder list of a .7z archive: (list of folder dicts, list of (size, crc or None) per file, total
    decoded size).  ctx is needed when the archive's header is itself compressed (7-Zip's default)."""
    buf = ctypes.create_string_buffer(data, len(data)) if len(data) else ctypes.create_string_buffer(1)
    h = ctx._h if ctx is not None else None
    nf, ns, total = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_uint64()
    st = N.lib().xlz_7z_index(h, ctypes.cast(buf, ctypes.c_void_p), len(data), None, 0, ctypes.byref(nf), None, 0,
                              ctypes.byref(ns), ctypes.byref(total))
    if st != OK:
        raise LzmaError(st, "xlz_7z_index")
    fo = (N.SzFolder * max(nf.value, 1))()
    su = (N.SzSubstream * max(ns.value, 1))()
    st = N.lib().xlz_7z_index(h, ctypes.cast(buf, ctypes.c_void_p), len(data), fo, nf.value, ctypes.byref(nf), su, ns.value,
                              ctypes.byref(ns), ctypes.byref(total))
    if st != OK:
        raise LzmaError(st, "xlz_7z_index")
    fields = [f for f, _ in N.SzFolder._fields_ if f != "reserved"]
    return ([{f: getattr(fo[i], f) for f in fields} for i in range(nf.value)],
            [(su[i].size, su[i].crc if su[i].has_crc else None) for i in range(ns.value)], total.value)


def sevenzip_decode(ctx, data, verify=True, max_size=None):
    """Decode every folder of a .7z archive as one GPU batch -> the files' bytes back to back.
    max_size: refuse (ERR_OUT_CAP) an archive whose header announces more decoded bytes than that -- the sizes come
    from an untrusted header and the output buffer is allocated from them."""
    _, _, total = sevenzip_index(data, ctx)
    if max_size is not None and total > max_size:
        raise LzmaError(ERR_OUT_CAP, "xlz_7z_decode: the archive announces %d bytes, max_size is %d" % (total, max_size))
    buf = ctypes.create_string_buffer(data, len(data))
    out = ctypes.create_string_buffer(max(total, 1))
    out_len = ctypes.c_uint64()
    unverified = ctypes.c_size_t()
    st = N.lib().xlz_7z_decode(ctx._h, ctypes.cast(buf, ctypes.c_void_p), len(data), ctypes.cast(out, ctypes.c_void_p),
                               total, ctypes.byref(out_len), 1 if verify else 0, ctypes.byref(unverified))
    if st != OK:
        raise LzmaError(st, "xlz_7z_decode")
    return out.raw[: out_len.value]
