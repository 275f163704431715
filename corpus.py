"""Synthetic corpora for tests/ and bench.py (test infrastructure, not product).

Plaintext is generated from a seed and compressed with the Python stdlib `lzma`
module (liblzma): because LZMA decoding of a valid stream is deterministic, the
plaintext itself is the expected output of the reference ("bit-exact vs the Go
reference" == "equals the plaintext").

Families:  T text-like, R random, M mixed 64 KiB segments,
Z long repeats.
"""
import hashlib
import lzma
import os
import struct
from concurrent.futures import ProcessPoolExecutor

import numpy as np

_VOCAB = None
_VOCAB_FLAT = None
_PIECE = 1 << 20


def _vocab():
    global _VOCAB
    if _VOCAB is None:
        rng = np.random.default_rng(20251003)
        _VOCAB = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 11))).astype(np.uint8))
                  for _ in range(2000)]
    return _VOCAB


def _vocab_flat():
    """(all words, each followed by a space, as one uint8 array; length of word+space; start)"""
    global _VOCAB_FLAT
    if _VOCAB_FLAT is None:
        v = _vocab()
        flat = np.frombuffer(b"".join(w + b" " for w in v), dtype=np.uint8)
        lens = np.array([len(w) + 1 for w in v], dtype=np.int32)
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        _VOCAB_FLAT = (flat, lens, starts)
    return _VOCAB_FLAT


def plain_text(seed, n):
    """Words drawn uniformly from a fixed 2000-word vocabulary, space separated
    (b" ".join(words)[:n], built with numpy index arithmetic instead of a Python join).
    Above 1 MiB the text is made of independent 1 MiB pieces (small working arrays stay in
    cache and in the allocator's arenas: several times faster than one big pass)."""
    if n > _PIECE:
        return b"".join(plain_text([seed, k], min(_PIECE, n - k * _PIECE)) for k in range((n + _PIECE - 1) // _PIECE))
    flat, lens, starts = _vocab_flat()
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(lens), size=n // 4 + 16)
    ln = lens[idx]
    ends = np.cumsum(ln, dtype=np.int64)
    k = min(int(np.searchsorted(ends, n + 1, side="left")) + 1, len(idx))
    ln = ln[:k]
    total = int(ends[k - 1])
    assert total - 1 >= n
    src = np.repeat((starts[idx[:k]] - (ends[:k] - ln)).astype(np.int32), ln)
    src += np.arange(total, dtype=np.int32)
    return flat[src][:n].tobytes()


def plain_random(seed, n):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def plain_mixed(seed, n, seg=65536):
    parts = []
    k = 0
    while sum(len(p) for p in parts) < n:
        m = min(seg, n - sum(len(p) for p in parts))
        parts.append(plain_text(seed * 1000 + k, m) if k % 2 == 0 else plain_random(seed * 1000 + k, m))
        k += 1
    return b"".join(parts)


def plain_repeats(seed, n):
    """Long repeats with rare mutations: match / rep dominated."""
    rng = np.random.default_rng(seed)
    pat = rng.integers(0, 256, size=int(rng.integers(1, 1500)), dtype=np.uint8)
    reps = n // len(pat) + 1
    buf = np.tile(pat, reps)[:n].copy()
    nmut = max(1, n // 20000)
    pos = rng.integers(0, n, size=nmut)
    buf[pos] = rng.integers(0, 256, size=nmut, dtype=np.uint8)
    return buf.tobytes()


def plain_far(seed, n, far_lo=6 << 20, far_hi=(8 << 20) - 1024, every=65536, blob=256):
    """Text with long-range repeats: every `every` bytes a random `blob`-byte key, copied again
    far_lo..far_hi bytes later -- matches whose distance approaches an 8 MiB dictionary (the
    window-wrap variant of BASELINE config 5)."""
    buf = np.frombuffer(plain_text(seed, n), dtype=np.uint8).copy()
    rng = np.random.default_rng(seed ^ 0x5EED)
    for q in range(0, n - blob, every):
        key = rng.integers(0, 256, size=blob, dtype=np.uint8)
        buf[q:q + blob] = key
        d = int(rng.integers(far_lo, far_hi))
        if q + d + blob <= n and (q + d) % every >= blob:  # do not land on another key
            buf[q + d:q + d + blob] = key
    return buf.tobytes()


FAMILIES = {"T": plain_text, "R": plain_random, "M": plain_mixed, "Z": plain_repeats, "F": plain_far}


def plain(family, seed, n):
    return FAMILIES[family](seed, n)


def props_byte(lc, lp, pb):
    return (pb * 5 + lp) * 9 + lc


def _enc(preset):
    """preset: a liblzma preset number, or a dict of explicit encoder options (mode, mf, nice_len,
    depth) for corpora that must be generated quickly."""
    return dict(preset) if isinstance(preset, dict) else {"preset": preset}


def lzma1_filters(dict_size=65536, lc=3, lp=0, pb=2, preset=6):
    return [dict(id=lzma.FILTER_LZMA1, dict_size=dict_size, lc=lc, lp=lp, pb=pb, **_enc(preset))]


def compress_alone(data, dict_size=65536, lc=3, lp=0, pb=2, preset=6, known_size=False):
    """.lzma (LZMA-alone) stream.  liblzma writes size=unknown + end marker; with
    known_size the 8-byte size field is patched (the `a_eos_and_size.lzma` flavour
    the reference's tests accept, reader1_test.go:38-43)."""
    c = lzma.compress(data, format=lzma.FORMAT_ALONE, filters=lzma1_filters(dict_size, lc, lp, pb, preset))
    if known_size:
        c = c[:5] + struct.pack("<Q", len(data)) + c[13:]
    return c


def compress_raw_lzma1(data, dict_size=65536, lc=3, lp=0, pb=2, preset=6):
    """Headerless LZMA1 payload (what a 7z folder holds) -> (props, dict_size, payload)."""
    c = lzma.compress(data, format=lzma.FORMAT_RAW, filters=lzma1_filters(dict_size, lc, lp, pb, preset))
    return props_byte(lc, lp, pb), dict_size, c


def compress_raw_lzma2(data, dict_size=65536, lc=3, lp=0, pb=2, preset=6):
    f = [dict(id=lzma.FILTER_LZMA2, dict_size=dict_size, lc=lc, lp=lp, pb=pb, **_enc(preset))]
    return lzma.compress(data, format=lzma.FORMAT_RAW, filters=f)


def lzma2_concat(segments, **kw):
    """One raw LZMA2 stream made of independently compressed segments: every segment
    starts with a dictionary reset (control 0xE0 or 0x01), i.e. K independent units
    (SURVEY.md §8d cfg4)."""
    parts = []
    for seg in segments:
        c = compress_raw_lzma2(seg, **kw)
        assert c[-1] == 0
        parts.append(c[:-1])
    return b"".join(parts) + b"\x00"


def alone_known_size_no_eos(data, dict_size=65536, lc=3, lp=0, pb=2, preset=6):
    """`a.lzma` flavour: size in the header, NO end marker.  Built from a single-chunk
    LZMA2 stream (plaintext <= 2 MiB, compressed <= 64 KiB); returns None when liblzma
    needed more than one chunk."""
    c = compress_raw_lzma2(data, dict_size, lc, lp, pb, preset)
    if not c or c[0] < 0xE0:
        return None
    unc = (((c[0] & 0x1F) << 16) | (c[1] << 8) | c[2]) + 1
    comp = ((c[3] << 8) | c[4]) + 1
    if unc != len(data) or 6 + comp + 1 != len(c) or c[-1] != 0:
        return None
    return bytes([c[5]]) + struct.pack("<I", dict_size) + struct.pack("<Q", unc) + c[6:6 + comp]


def _make_one(args):
    family, seed, n, kw = args
    p = plain(family, seed, n)
    c = compress_alone(p, **kw)
    return c, hashlib.sha256(p).digest()


def make_alone_batch(family, n_streams, size, base_seed=1, workers=None, **kw):
    """n_streams independent .lzma streams of `size` plaintext bytes each.
    Returns (list of compressed bytes, list of sha256 digests of the plaintext)."""
    jobs = [(family, base_seed + i, size, kw) for i in range(n_streams)]
    workers = workers or min(os.cpu_count() or 1, 32)
    if workers <= 1 or n_streams < 4:
        res = [_make_one(j) for j in jobs]
    else:
        with ProcessPoolExecutor(max_workers=workers) as ex:
            res = list(ex.map(_make_one, jobs, chunksize=max(1, n_streams // (workers * 8))))
    return [r[0] for r in res], [r[1] for r in res]


def _make_one_lzma2(args):
    family, seed, n_seg, seg_size, kw = args
    segs = [plain(family, seed * 4099 + k, seg_size) for k in range(n_seg)]
    h = hashlib.sha256()
    for sg in segs:
        h.update(sg)
    return lzma2_concat(segs, **kw), h.digest()


def make_lzma2_batch(family, n_streams, n_segments, seg_size, base_seed=1, workers=None, **kw):
    """n_streams raw LZMA2 streams, each n_segments independently compressed segments."""
    jobs = [(family, base_seed + i, n_segments, seg_size, kw) for i in range(n_streams)]
    workers = workers or min(os.cpu_count() or 1, 32)
    if n_streams >= 4 and workers > 1:
        with ProcessPoolExecutor(max_workers=workers) as ex:
            res = list(ex.map(_make_one_lzma2, jobs, chunksize=1))
    elif workers > 1 and n_segments >= 8:
        # few big streams: parallelise over segments instead
        res = []
        for (fam, seed, n_seg, seg_size_, kw_) in jobs:
            with ProcessPoolExecutor(max_workers=workers) as ex:
                parts = list(ex.map(_lzma2_segment, [(fam, seed * 4099 + k, seg_size_, kw_) for k in range(n_seg)],
                                    chunksize=max(1, n_seg // (workers * 4))))
            h = hashlib.sha256()
            for _, pl in parts:
                h.update(pl)
            res.append((b"".join(c for c, _ in parts) + b"\x00", h.digest()))
    else:
        res = [_make_one_lzma2(j) for j in jobs]
    return [r[0] for r in res], [r[1] for r in res]


def _lzma2_segment(args):
    family, seed, seg_size, kw = args
    p = plain(family, seed, seg_size)
    c = compress_raw_lzma2(p, **kw)
    assert c[-1] == 0
    return c[:-1], p
