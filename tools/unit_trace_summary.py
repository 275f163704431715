"""Markdown summary of per-unit time stamps saved by `bench.py --trace-out` (xlz_batch_unit_trace):
how long each unit occupied its wave, when it started, how that relates to the work-queue key.
usage: python tools/unit_trace_summary.py <title> <trace.npz> [<title> <trace.npz> ...]"""
import sys
import numpy as np

args = sys.argv[1:]
print("# Unit durations inside one launch of xlz_decode_kernel\n")
print("Per-unit stamps of the device's 100 MHz clock (`UnitResult.t_start/t_end`), last timed launch of "
      "`bench.py --trace-out`.  Slots = resident single-wave workgroups (4096).\n")
for title, fn in zip(args[0::2], args[1::2]):
    d = np.load(fn)
    t0 = d["t_start"].astype(np.int64)
    t1 = d["t_end"].astype(np.int64)
    il = d["in_len"].astype(np.float64)
    dur = (t1 - t0) / 1e5
    span = (t1.max() - t0.min()) / 1e5
    slots = min(len(dur), 4096)
    print("## %s\n" % title)
    print("- units %d, launch span %.1f ms, slot occupancy %.3f (sum of unit durations / (slots x span))" %
          (len(dur), span, dur.sum() / (slots * span)))
    print("- unit duration ms: min %.1f, p10 %.1f, p50 %.1f, p90 %.1f, max %.1f" % tuple(np.percentile(dur, [0, 10, 50, 90, 100])))
    print("- correlation of the work-queue key (compressed bytes of the unit) with the duration: %.3f" % np.corrcoef(il, dur)[0, 1])
    h, edges = np.histogram(dur, bins=16)
    print("\n| duration bin (ms) | units |\n|---|---|")
    for k in range(len(h)):
        print("| %.1f - %.1f | %d |" % (edges[k], edges[k + 1], h[k]))
    print()
