"""Differential fuzz of the HIP path against the CPU oracle (dev tool, run on the GPU box):
random families / sizes / lc-lp-pb / dictionary sizes (odd ones included), known-size headers,
truncations, byte flips, LZMA2 concatenations with damaged framing.  Every stream is compared on
bytes, status and consumed input.
usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import os, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import corpus, lzma_amd, oracle
from lzma_amd import Stream, FMT_LZMA_ALONE, FMT_LZMA2_RAW
from lzma_craft import ANY_PROPS, SMALL_PROPS, random_lzma2_stream  # packet-level crafter: LZMA2 streams with every chunk / reset kind

def damage(rng, c):
    c = bytearray(c)
    k = int(rng.integers(0, 4))
    if k == 0 and len(c) > 20:
        return bytes(c[: int(rng.integers(14, len(c)))])
    if k == 1 and len(c) > 20:
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(13, len(c)))] ^= 1 << int(rng.integers(0, 8))
        return bytes(c)
    if k == 2 and len(c) > 40:
        a = int(rng.integers(13, len(c) - 8))
        c[a:a + 4] = rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
        return bytes(c)
    return bytes(c)


def fuzz(ctx, budget, seed, per_round=160, verbose=True):
  """-> (streams compared, streams with a non-OK status); raises AssertionError on a mismatch"""
  rng = np.random.default_rng(seed)
  t_end = time.time() + budget
  n_total = n_bad_status = 0
  rounds = n_sliced_rounds = 0
  while time.time() < t_end:
      rounds += 1
      streams, wants = [], []
      # half of the rounds keep every pb <= 2, so that the launch uses the COMPACT model layout (round 5: room for four
      # posStates, 24 workgroups per CU); the other half mixes pb up to 4 in: the full layout
      pb_max = 2 if rounds % 4 < 2 else 4
      # ... and every other compact round runs the loop with BRANCHY decisions (what launches of 24 workgroups per CU run;
      # XLZ_BRANCHY is the library's development switch for it)
      os.environ["XLZ_BRANCHY"] = "1" if rounds % 4 == 1 else "0"
      for _ in range(per_round):
          fam = "TRMZ"[int(rng.integers(0, 4))]
          n = int(rng.choice([1, 2, 17, 300, 335, 336, 337, 1000, 5000, 40000, 70000, 131072, 200000, 300001]))
          if rng.random() < 0.02:
              n = int(rng.choice([1 << 20, 3 << 20]))  # rare: streams far longer than the small dictionaries
          n = max(1, n + int(rng.integers(-3, 4)))
          lc = int(rng.integers(0, 5)); lp = int(rng.integers(0, 5 - lc)); pb = int(rng.integers(0, pb_max + 1))
          dict_size = int(rng.choice([4096, 4097, 5000, 8192, 65536, 65537, 100003, 1 << 20, 8 << 20]))
          p = corpus.plain(fam, int(rng.integers(1, 1 << 30)), n)
          if rng.random() < 0.75:
              c = corpus.compress_alone(p, dict_size=dict_size, lc=lc, lp=lp, pb=pb, known_size=bool(rng.random() < 0.3))
              cap = n + int(rng.choice([0, 0, 0, 1, 100, -1, -50])) if n > 60 else n
              cap = max(cap, 0)
              if rng.random() < 0.35:
                  c = damage(rng, c)
              streams.append(Stream(c, FMT_LZMA_ALONE, out_cap=cap))
              wants.append(oracle.lzma1_alone(c, cap))
          else:
              segs, k = [], 0
              nseg = int(rng.integers(1, 5))
              cut = sorted(set(int(x) for x in rng.integers(0, n + 1, nseg - 1)))
              parts = [p[a:b] for a, b in zip([0] + cut, cut + [n]) if b > a] or [p]
              d2 = int(rng.choice([4096, 65536, 1 << 20, 8 << 20]))
              c = corpus.lzma2_concat(parts, dict_size=d2, lc=lc, lp=lp, pb=pb)
              if rng.random() < 0.4:
                  c = bytearray(c)
                  for _ in range(int(rng.integers(1, 3))):
                      c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
                  c = bytes(c)
              cap = n + int(rng.choice([0, 0, 5, -1]))
              cap = max(cap, 0)
              streams.append(Stream(c, FMT_LZMA2_RAW, out_cap=cap, dict_size=d2))
              wants.append(oracle.lzma2_raw(c, d2, cap))
      # crafted LZMA2 streams: random packets, random resets; rep matches may read the window bytes an
      # EARLIER dictionary epoch left behind (window.go:135-140) -- the exact (epoch table) launch
      for _ in range(per_round // 6):
          r2 = random.Random(int(rng.integers(1, 1 << 62)))
          d2 = r2.choice([4096, 4097, 8192, 65536])
          wide = ANY_PROPS if pb_max == 4 else [q for q in ANY_PROPS if q[2] <= 2 or q[0] + q[1] > 8]   # (lc+lp > 8: the HBM-model launch)
          c, expect = random_lzma2_stream(r2, d2, props=wide if r2.random() < 0.3 else SMALL_PROPS)
          cap = len(expect) + int(rng.choice([0, 0, 7, -1])) if len(expect) > 1 else len(expect)
          if rng.random() < 0.2 and len(c) > 8:
              c = bytearray(c)
              c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
              c = bytes(c)
          streams.append(Stream(c, FMT_LZMA2_RAW, out_cap=max(cap, 0), dict_size=d2))
          wants.append(oracle.lzma2_raw(c, d2, max(cap, 0)))
      # every other round through the SLICED form of the call (round 5: a sequence of launches, heads of the inputs first,
      # slices downloaded under the decode) with a random number of slices; the results must not depend on it
      k_slices = int(rng.choice([2, 3, 5, 8, 13])) if rounds % 2 else 1
      ctx.set_slicing(1 if k_slices > 1 else 0, 1 << 16, k_slices)
      got = lzma_amd.decode_batch(ctx, streams)
      if k_slices > 1:
          st = ctx.last_call_stats()
          n_sliced_rounds += 1 if st["slices"] > 1 else 0
      for i, (g, w) in enumerate(zip(got, wants)):
          n_total += 1
          if g[1] != 0:
              n_bad_status += 1
          if g != w:
              s = streams[i]
              fn = "gpurun_out/fuzz_fail_%d_%d.bin" % (seed, n_total)
              os.makedirs("gpurun_out", exist_ok=True)
              open(fn, "wb").write(s.data)
              d = next((k for k in range(min(len(g[0]), len(w[0]))) if g[0][k] != w[0][k]), None)
              print("MISMATCH stream %d fmt %d cap %d dict %d: gpu (st %d, len %d, in %d) oracle (st %d, len %d, in %d) first diff %s -> %s"
                    % (i, s.fmt, s.out_cap, s.dict_size, g[1], len(g[0]), g[2], w[1], len(w[0]), w[2], d, fn), flush=True)
              raise AssertionError("GPU and oracle differ, input saved as " + fn)
      if verbose:
        print("round %d: %d streams ok so far (%d with a non-OK status; %d rounds ran in slices)" % (rounds, n_total, n_bad_status, n_sliced_rounds), flush=True)
  ctx.set_slicing(0, 0, 0)
  os.environ.pop("XLZ_BRANCHY", None)
  return n_total, n_bad_status


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n, nb = fuzz(lzma_amd.Context(0), budget, seed)
    print("fuzz ok: %d streams (%d with a non-OK status)" % (n, nb))
