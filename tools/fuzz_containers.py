"""Mutation fuzz of the container front-ends WITH decode (dev tool, run on the GPU box): .xz files against
liblzma (Python's lzma module), hand-built .7z archives against their own plaintext.
  .xz: whatever liblzma decodes, xlz_xz_decode must decode to the same bytes; where liblzma refuses the file,
       xlz_xz_decode (verify on) must refuse it too -- unless liblzma's refusal arises INSIDE a block's LZMA2 payload:
       there the front-end is as lenient as the reference's own LZMA2 reader is (reader2.go:185-198, parity note 7).
       Every such input is classified by the place of liblzma's refusal (classify_lenient), saved under
       gpurun_out/lenient_xz_<seed>/, and anything but a payload case fails the run.
  .7z: a decode with verification on that succeeds has passed every CRC the archive carries: its bytes must be
       the original files'.
usage: python tools/fuzz_containers.py [seconds] [seed]"""
import lzma
import os
import random
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus
import lzma_amd
from fuzz_parsers import _mutate
from sevenzip_craft import archive, copy_folder, lzma2_folder, lzma_folder

def strict_xz(b):
    """liblzma, stream by stream (lzma.decompress ignores whatever follows the first stream once that one is
    good, and knows no stream padding): every stream must decode, padding is 4-byte groups of zeros"""
    out, data = [], b
    if not data:
        return None
    while data:
        d = lzma.LZMADecompressor(format=lzma.FORMAT_XZ)
        try:
            out.append(d.decompress(data))
        except lzma.LZMAError:
            return None
        if not d.eof:
            return None
        data = d.unused_data
        k = 0
        while data[k:k + 4] == b"\0\0\0\0":
            k += 4
        data = data[k:]
    return b"".join(out)


LIMIT = 64 << 20  # never allocate more than this for an output the (mutated) index announces

CHECK_SIZE = {0: 0, 1: 4, 4: 8, 10: 32}
# where liblzma's refusal of an .xz file arises: PAYLOAD is the reference's own leniency (its LZMA2 reader takes control bytes
# 0x03..0x7f as a clean end, reader2.go:185-198, parity note 7; it does not refuse a chunk that ends short of its announced
# sizes, ...): the container front-end hands the block's payload to exactly that reader, DESIGN.md section 5.  Every other place
# is container structure that xlz_xz_index / xlz_xz_decode must check as liblzma does: a lenient case there is a bug.
PAYLOAD, CONTAINER = "payload (the reference's LZMA2 leniency)", "CONTAINER"


def liblzma_error_offset(b):
    """the input offset at which liblzma (stream by stream, fed one byte at a time) refuses the file; len(b) if it only
    runs out of input"""
    i = 0
    while i < len(b):
        d = lzma.LZMADecompressor(format=lzma.FORMAT_XZ)
        while not d.eof:
            if i == len(b):
                return i
            try:
                d.decompress(b[i:i + 1], 1 << 16)
                while not d.needs_input and not d.eof:
                    d.decompress(b"", 1 << 16)
            except lzma.LZMAError:
                return i
            i += 1
        while b[i:i + 4] == b"\0\0\0\0":
            i += 4
    return None


def classify_lenient(b):
    """-> (bucket, detail) for an .xz input that liblzma refuses and xlz_xz_decode accepted"""
    off = liblzma_error_offset(b)
    if off is None:
        return CONTAINER, "liblzma refuses the file as a whole but not while reading it (stream padding?)"
    blocks, _ = lzma_amd.xz_index(b)
    for k, bl in enumerate(blocks):
        pay0, pay1 = bl["comp_off"], bl["comp_off"] + bl["comp_len"]
        chk1 = bl["check_off"] + CHECK_SIZE.get(bl["check_type"], 0)
        if pay0 <= off < pay1 or (off == pay1 and bl["comp_len"] == 0):
            return PAYLOAD, "block %d, payload offset %d of %d" % (k, off - pay0, bl["comp_len"])
        if pay1 <= off < bl["check_off"]:
            # liblzma notices a payload that ends early (or runs on) where the NEXT field begins: decode the payload alone
            f = [{"id": lzma.FILTER_LZMA2, "dict_size": max(bl["dict_size"], 4096)}]
            try:
                out = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=f).decompress(b[pay0:pay1])
                if len(out) == bl["uncomp_len"]:
                    return CONTAINER, "block %d padding at %d" % (k, off)
            except lzma.LZMAError:
                pass
            return PAYLOAD, "block %d: liblzma's LZMA2 decoder refuses the payload (noticed at the padding, offset %d)" % (k, off)
        if bl["check_off"] <= off < chk1:
            f = [{"id": lzma.FILTER_LZMA2, "dict_size": max(bl["dict_size"], 4096)}]
            try:
                d = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=f)
                out = d.decompress(b[pay0:pay1])
                if len(out) == bl["uncomp_len"] and d.eof:
                    return CONTAINER, "block %d check field at %d" % (k, off)
            except lzma.LZMAError:
                pass
            return PAYLOAD, "block %d: payload differs for liblzma (noticed at the check, offset %d)" % (k, off)
    return CONTAINER, "offset %d of %d: stream header / block header / index / footer / stream padding" % (off, len(b))


def fuzz(ctx, seconds, seed, verbose=True):
    rnd = random.Random(seed)
    plains = [corpus.plain("TRMZ"[i % 4], 4000 + i, n) for i, n in enumerate((1, 300, 5000, 70000, 300000))]
    xzs = []
    for p in plains:
        for chk in (lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256):
            xzs.append(lzma.compress(p, format=lzma.FORMAT_XZ, check=chk, preset=rnd.choice([0, 1, 6])))
    xzs.append(xzs[5] + xzs[10])
    xzs.append(xzs[2] + b"\0" * 4 + xzs[14] + xzs[7])
    szs = []
    for combo in ([0], [0, 1], [1, 2, 3]):
        fo, want = [], b""
        for c in combo:
            parts = plains[: 1 + c]
            data = b"".join(parts)
            rec, packed = (lzma_folder, lzma2_folder, copy_folder, lzma_folder)[c](data)
            fo.append((rec, packed, parts))
            want += data
        for fc in (True, False):
            a = archive(fo, folder_crc=fc)
            szs.append((a, 32 + struct.unpack("<Q", a[12:20])[0], want))
    t_end = time.time() + seconds
    n = n_ok = n_lenient = 0
    buckets = {}
    save_dir = os.path.join(ROOT, "gpurun_out", "lenient_xz_%d" % seed)
    while time.time() < t_end:
        n += 1
        if rnd.random() < 0.6:
            b = bytes(_mutate(rnd, rnd.choice(xzs))) if rnd.random() < 0.85 else rnd.choice(xzs)
            ref = strict_xz(b)
            try:
                _, total = lzma_amd.xz_index(b)
                if total > LIMIT:
                    continue
                got = lzma_amd.xz_decode(ctx, b, verify=True)
            except lzma_amd.LzmaError:
                got = None
            if ref is not None:
                assert got == ref, "xz: liblzma decodes %d bytes, xlz %s" % (len(ref), None if got is None else len(got))
                n_ok += 1
            elif got is not None:
                # liblzma refuses, xlz decodes: say WHERE liblzma's refusal arises and keep the input
                n_lenient += 1
                bucket, detail = classify_lenient(b)
                buckets[bucket] = buckets.get(bucket, 0) + 1
                os.makedirs(save_dir, exist_ok=True)
                name = "%s_%d.xz" % ("payload" if bucket == PAYLOAD else "container", n)
                open(os.path.join(save_dir, name), "wb").write(b)
                if verbose:
                    print("lenient input %d (%s): %s: %s" % (n, name, bucket, detail), flush=True)
                assert bucket == PAYLOAD, "an .xz file with damaged CONTAINER structure was accepted: %s (%s)" % (detail, name)
        else:
            a, hs, want = rnd.choice(szs)
            b = _mutate(rnd, a, tail_from=hs if rnd.random() < 0.5 else 32)
            if len(b) > max(hs, 32) and rnd.random() < 0.7:
                b[28:32] = struct.pack("<I", zlib.crc32(bytes(b[hs:])) & 0xFFFFFFFF)
                b[8:12] = struct.pack("<I", zlib.crc32(bytes(b[12:32])) & 0xFFFFFFFF)
            b = bytes(b)
            try:
                folders, subs, total = lzma_amd.sevenzip_index(b, ctx)
                if total > LIMIT:
                    continue
                got = lzma_amd.sevenzip_decode(ctx, b, verify=True)
            except lzma_amd.LzmaError:
                continue
            # every folder (or file) of these archives carries a CRC: what passed them is the plaintext, unless the
            # mutation dropped folders / files from the header itself -- or the CRC definitions (round 3, seed 3304
            # input 88 666: a header mutated into one without any digest + one flipped data byte: an UNVERIFIED decode,
            # which proves nothing)
            covered = all(f["has_crc"] or all(subs[f["first_substream"] + k][1] is not None for k in range(f["n_substreams"]))
                          for f in folders)
            if len(got) == len(want) and covered:
                assert got == want, "7z: verified decode differs from the plaintext"
            n_ok += 1
        if verbose and n % 200 == 0:
            print("%d inputs, %d decoded and equal, %d xz files liblzma refuses but xlz decodes" % (n, n_ok, n_lenient), flush=True)
    if verbose:
        print("lenient .xz inputs by where liblzma refuses them: %r" % (buckets,), flush=True)
    return n, n_ok, n_lenient


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("container fuzz ok: %d inputs, %d decoded and equal, %d lenient" % fuzz(lzma_amd.Context(0), secs, seed))
