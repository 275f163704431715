"""Regenerates tests/golden/expected.json from the oracle (run from the repo root).

The nine data files are the reference's own test assets (testassets/*.lzma*);
the two MD5s are the only outputs the reference pins (reader1_test.go:107).
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle  # noqa: E402

G = os.path.dirname(os.path.abspath(__file__))
ALONE = ["a.lzma", "a_eos.lzma", "a_eos_and_size.lzma", "a_lp1_lc2_pb1.lzma", "bad_corrupted.lzma",
         "bad_eos_incorrect_size.lzma", "bad_incorrect_size.lzma", "randomfile.dat.lzma"]


def entry(out, st, ic, **kw):
    e = {"status": st, "out_len": len(out), "in_consumed": ic,
         "sha256": hashlib.sha256(out).hexdigest(), "md5": hashlib.md5(out).hexdigest()}
    e.update(kw)
    return e


if __name__ == "__main__":
    old = json.load(open(os.path.join(G, "expected.json")))
    files = {}
    for f in ALONE:
        d = open(os.path.join(G, f), "rb").read()
        files[f] = entry(*oracle.lzma1_alone(d, 2 << 20), format="alone")
    d = open(os.path.join(G, "randomfile.dat.lzma2"), "rb").read()
    files["randomfile.dat.lzma2"] = entry(*oracle.lzma2_raw(d, 0, 2 << 20), format="lzma2", dict_size=0)
    json.dump({"notes": old["notes"], "files": files}, open(os.path.join(G, "expected.json"), "w"),
              indent=1, sort_keys=True)
