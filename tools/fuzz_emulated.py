"""Differential fuzz of the GENERATED FAST LOOP on the CPU: random liblzma streams (families, sizes, lc/lp/pb,
dictionary sizes incl. odd ones that wrap, known / unknown size, an output offset as behind an LZMA2 dictionary
reset) decoded by tools/gcn_emu.py running the loop -- the committed one or a set of generator switches -- and
compared with the plaintext and, where the loop hands over, with the Python restatement's range / code / state /
reps / prevByte / input position.  No GPU.
usage: python tools/fuzz_emulated.py [seconds] [seed] [--variant a,b] [--without a,b] [--strict]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus
import oracle
import test_fastpath_emulated as T


def main():
    a = sys.argv[1:]
    add = a[a.index("--variant") + 1].split(",") if "--variant" in a else []
    rem = a[a.index("--without") + 1].split(",") if "--without" in a else []
    pos = [x for i, x in enumerate(a) if not x.startswith("--") and (i == 0 or a[i - 1] not in ("--variant", "--without"))]
    secs = float(pos[0]) if pos else 60.0
    rnd = random.Random(int(pos[1]) if len(pos) > 1 else 1)
    prog = T._render(add, rem)
    t_end = time.time() + secs
    n = nbytes = 0
    while time.time() < t_end:
        fam = rnd.choice("TRMZ")
        size = rnd.choice([400, 900, 2000, 3500, 6000])
        lc = rnd.randrange(0, 5)
        lp = rnd.randrange(0, 5 - lc)
        pb = rnd.randrange(0, 5)
        ds = rnd.choice([4096, 4097, 5000, 6145, 8192, 65536])
        base = rnd.choice([0, 0, 0, 1, 4097, 70001])
        known = rnd.random() < 0.6
        p = corpus.plain(fam, rnd.randrange(1, 1 << 30), size)
        blob = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=known, preset=rnd.choice([0, 6]))
        if rnd.random() < 0.3:  # a header that claims a smaller, odd dictionary: wraps where the encoder saw none, posState
            blob = blob[:1] + rnd.choice([4097, 4099, 5001]).to_bytes(4, "little") + blob[5:]  # follows the WRAPPED position
        ds = max(4096, int.from_bytes(blob[1:5], "little"))  # what the header says (liblzma rounds the size up), reader1.go:199-201
        junk = corpus.plain("R", 5, base) if base else b""
        what = "family %s size %d lc%d lp%d pb%d dict %d base %d known %s" % (fam, size, lc, lp, pb, ds, base, known)
        # what the REFERENCE makes of the stream: with a dictionary size that is no multiple of 2^pb / 2^lp its
        # posState follows the WRAPPED window position (decompress.go:22), so behind the first wrap it decodes
        # something else than the plaintext, possibly into an error -- the oracle says what
        want, status, _ = oracle.lzma1_alone(blob, size + 8192)
        try:
            cap = base + size + (0 if known else 4096)
            # (64 bytes behind a stream that ends in its marker let the loop itself meet the marker; a derailed decode
            # must stop where the real input ends: the kernel hands the last 32 bytes to the checked path)
            pad = b"\0" * 64 if (not known and status == 0 and want == p) else b""
            out, m, entries, exits, in_pos = T.run_fast_loop(prog, blob[13:] + pad, lc, lp, pb, ds, cap,
                                                            junk + want + b"\0" * 8192, strict_waits="--strict" in a,
                                                            dpp="hdpp" not in rem, base=base)
            assert out == want[:len(out)], "bytes differ"
            if exits[1]:
                assert status == oracle.ERR_RESULT and out == want, "error exit where the oracle has none"
            elif exits[2]:
                assert out == want and ((m.s["code"] == 0) == (status >= 0)), "end marker exit"
            else:
                rc, st = T._reference_state_at(blob[13:], lc, lp, pb, ds, size if known else None, len(out))
                got = (m.s["range"], m.s["code"], m.s["state"], [m.s["rep0"], m.s["rep1"], m.s["rep2"], m.s["rep3"]], in_pos)
                assert got == (rc.range, rc.code, st.state, st.reps, rc.p), "state differs at the hand-over: %r" % (got,)
                if out:
                    assert m.s["prev"] == want[len(out) - 1], "prevByte differs"
        except Exception as e:  # noqa: BLE001
            print("MISMATCH (%s): %s: %s" % (what, type(e).__name__, e), flush=True)
            raise
        n += 1
        nbytes += len(out)
        if n % 20 == 0:
            print("%d streams, %d bytes decoded by the emulated loop" % (n, nbytes), flush=True)
    print("emulated fuzz ok: %d streams, %d bytes" % (n, nbytes))


if __name__ == "__main__":
    main()
