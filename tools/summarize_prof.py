"""Condense the rocprofv3 output of tools/profile_bench.sh into a small markdown summary."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]


def kernel_rows(sub, pattern):
    f = glob.glob(os.path.join(O, sub, "*", pattern))
    return list(csv.DictReader(open(f[0]))) if f else []


print("# rocprofv3 summary (%s)\n" % os.path.basename(O))
print("command: `python3 bench.py %s`\n" % open(os.path.join(O, "command.txt")).read().strip().replace("bench args: ", ""))
for name in ("kt", "fetch", "write", "sq", "sq2"):
    p = os.path.join(O, name + ".json")
    if os.path.exists(p) and os.path.getsize(p):
        j = json.load(open(p))   # bench.py --detail-out
        print("- %s pass bench line: value %.3f %s, roofline.kernel_ms %.3f, achieved %.3f GB/s" % (
            name, j["value"], j["unit"], j["roofline"]["kernel_ms"], j["roofline"]["achieved"]))
        algo = j["roofline"]["algorithmic_bytes_per_launch"]
print("\n## --kernel-trace --stats\n")
print("| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|")
for r in kernel_rows("kt", "*kernel_stats.csv"):
    print("| %s | %s | %s | %.0f | %s |" % (r["Name"][:60], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
tr = kernel_rows("kt", "*kernel_trace.csv")
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr if "xlz_decode" in r["Kernel_Name"]]
if d:
    print("\nper launch of the decode kernel, in launch order (ms): %s -- the first is bench.py's untimed warm-up "
          "(cold clocks / caches), the rest are the timed steps and agree with the HIP-event figure above; the "
          "`--stats` average is over all of them." % ", ".join("%.2f" % x for x in d))
print("\n## PMC (per launch of xlz_decode_kernel, summed over the device)\n")
vals = collections.defaultdict(list)
for sub in ("fetch", "write", "sq", "sq2", "lat"):
    for r in kernel_rows(sub, "*counter_collection.csv"):
        if "xlz_decode" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| counter | per launch (mean) | launches |\n|---|---|---|")
for k, v in sorted(vals.items()):
    print("| %s | %.6g | %d |" % (k, sum(v) / len(v), len(v)))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    f = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024
    w = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
    print("\nHBM traffic per launch: FETCH_SIZE %.3f GB (raw, KiB units x1024; gfx950 reports half of wide streaming "
          "reads -- this kernel's reads are byte/dword gathers, uncalibrated), WRITE_SIZE %.3f GB; algorithmic bytes "
          "per launch %.3f GB." % (f / 1e9, w / 1e9, algo / 1e9))
