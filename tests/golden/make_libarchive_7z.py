"""Generates tests/golden/libarchive_solid.7z and its expected index libarchive_solid.json: an archive written by an
INDEPENDENT writer -- libarchive's 7zip writer, through `cmake -E tar cf x.7z --format=7zip` (the image has no 7-Zip,
but it has this) -- with a solid LZMA1 folder, an LZMA-encoded header and per-file CRCs.  The expected values come from
the files themselves (sizes, zlib CRCs, SHA-256 of their bytes in archive order) and from tests/sevenzip_read.py, a
plain-Python reader that is independent of the product's parser; liblzma decodes the folder as a cross-check.
    python tests/golden/make_libarchive_7z.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import corpus          # noqa: E402
import sevenzip_read   # noqa: E402


def write_archive(files, path):
    """files: [(name, bytes)] -> the archive's bytes (cmake / libarchive writes the entries in the order given)"""
    with tempfile.TemporaryDirectory() as d:
        for name, data in files:
            with open(os.path.join(d, name), "wb") as f:
                f.write(data)
        subprocess.check_call(["cmake", "-E", "tar", "cf", path, "--format=7zip"] + [n for n, _ in files], cwd=d)
    return open(path, "rb").read()


def main():
    files = [("words.txt", corpus.plain("T", 7001, 30_000)), ("noise.bin", corpus.plain("R", 7002, 4_000)),
             ("empty.dat", b""), ("repeats.txt", corpus.plain("Z", 7003, 20_000)), ("mixed.bin", corpus.plain("M", 7004, 9_000))]
    out = os.path.join(HERE, "libarchive_solid.7z")
    data = write_archive(files, out)
    r = sevenzip_read.read(data)
    m = r["main"]
    assert r["encoded"] and len(m["folders"]) == 1
    f = m["folders"][0]
    streams = [(len(b), zlib.crc32(b)) for _, b in files if b]   # an empty file has no stream
    assert [tuple(x) for x in m["substreams"][0]] == streams, (m["substreams"], streams)
    content = b"".join(b for _, b in files)
    packed = data[32 + m["pack_pos"]:32 + m["pack_pos"] + m["pack_sizes"][0]]
    assert sevenzip_read.lzma1_decode(packed, f["props"], f["unpack_size"]) == content
    exp = {"writer": subprocess.check_output(["cmake", "--version"], text=True).splitlines()[0] + " (bundled libarchive), cmake -E tar cf --format=7zip",
           "files": [[n, len(b)] for n, b in files], "encoded_header": True,
           "folder": {"method": f["method"].hex(), "props": f["props"].hex(), "unpack_size": f["unpack_size"],
                      "pack_off": 32 + m["pack_pos"], "pack_len": m["pack_sizes"][0]},
           "substreams": [list(x) for x in streams], "sha256": hashlib.sha256(content).hexdigest(), "archive_bytes": len(data)}
    with open(os.path.join(HERE, "libarchive_solid.json"), "w") as fh:
        json.dump(exp, fh, indent=1)
    print("wrote %s (%d bytes) and its index" % (out, len(data)))


if __name__ == "__main__":
    main()
