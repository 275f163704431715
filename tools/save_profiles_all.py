"""Cut a tools/profile_all.sh run (every bench.py config decoded by ONE process per rocprofv3 pass) into per-config
profile records under profiles/<round>/ and register them in profiles/current.json -- the same files
tools/save_profile.py writes for a one-config run: <cfg>_summary.md, _kernel_stats.csv, _kernel_trace.csv, _pmc.csv,
_benchline.json.
    python tools/save_profiles_all.py <tag> <dest-dir>          e.g.  r03all profiles/r03
How the dispatches are told apart: bench.py decodes the headline first and then the side configs in the order of its
command line, every leg is exactly (warm-up + steps) launches of xlz::xlz_decode_kernel; the number of dispatches in
every pass is checked against that before anything is written.  The kernel_stats of a side config are computed from
the kernel trace (rocprofv3's own --stats table covers the whole process: kept as all_kernel_stats.csv); the headline's
own --stats table comes from the extra one-config pass `kt_head` when the run has one."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import save_profile as SP

KERNELS = ("xlz::xlz_decode_kernel(", "xlz::xlz_decode_kernel_pb2(", "xlz::xlz_decode_kernel_pb2_br(")   # full layout; compact (pb <= 2); compact with branchy decisions (24 per CU)
WIDE_READS = {"cfg4-R"}   # configs whose launch is the stored-chunk copy: 16-byte-per-lane streaming reads (FETCH_SIZE x 2)


def parse_command(text):
    a = text.replace("bench args:", "").split()
    opt = lambda k, d=None: a[a.index(k) + 1] if k in a else d
    head = opt("--headline", "cfg3")
    side = [c for c in opt("--configs", "").split(",") if c and c not in ("none", head)]
    per_head = int(opt("--steps", 5)) + int(opt("--warmup", 1))
    per_side = int(opt("--side-steps", 3)) + 1
    return [(head, per_head)] + [(c, per_side) for c in side]


def split(items, legs, what):
    n = sum(k for _, k in legs)
    assert len(items) == n, "%s: %d launches of the decode kernel, the command line makes %d" % (what, len(items), n)
    out, i = {}, 0
    for name, k in legs:
        out[name] = items[i:i + k]
        i += k
    return out


def one(pattern):
    f = glob.glob(pattern)
    assert f, "missing " + pattern
    return f[0]


def main():
    import bench
    tag, dest = sys.argv[1], sys.argv[2]
    src = os.path.join("gpurun_out", "prof_" + tag)
    os.makedirs(dest, exist_ok=True)
    command = open(os.path.join(src, "command.txt")).read().strip()
    legs = parse_command(command)
    lines = {p: json.load(open(os.path.join(src, p + ".json"))) for p in ("kt", "fetch", "write", "sq", "sq2")}   # bench.py --detail-out: the full record
    line = lines["kt"]
    rev = line["config"]["kernel_rev"]
    assert rev == bench.kernel_rev(), "the profile was taken on another build than lzma_amd/libxlz.so"
    per_cfg_line = {legs[0][0]: line}
    for c in line.get("configs") or []:
        per_cfg_line[c["name"]] = c

    # ---- kernel trace: launches in time order
    tr = [r for r in csv.DictReader(open(one(src + "/kt/*/*kernel_trace.csv"))) if r["Kernel_Name"].startswith(KERNELS)]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    header = list(tr[0].keys())
    tr_by = split(tr, legs, "kernel trace")
    shutil.copy(one(src + "/kt/*/*kernel_stats.csv"), os.path.join(dest, "all_kernel_stats.csv"))

    # ---- counters: dispatch ids in launch order, pass by pass
    rows_by = collections.defaultdict(list)
    vals_by = collections.defaultdict(lambda: collections.defaultdict(list))
    grid_by = {}
    for sub in ("fetch", "write", "sq", "sq2"):
        rs = [r for r in csv.DictReader(open(one(src + "/%s/*/*counter_collection.csv" % sub))) if r["Kernel_Name"].startswith(KERNELS)]
        ids = sorted({int(r["Dispatch_Id"]) for r in rs})
        which = {}
        for name, chunk in split(ids, legs, sub + " pass").items():
            for d in chunk:
                which[d] = name
        for r in sorted(rs, key=lambda r: (int(r["Dispatch_Id"]), r["Counter_Name"])):
            name = which[int(r["Dispatch_Id"])]
            rows_by[name].append(SP.pmc_row(sub, r))
            vals_by[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid_by[name] = int(r["Grid_Size"])

    head_stats = glob.glob(src + "/kt_head/*/*kernel_stats.csv")
    for name, _ in legs:
        pre = os.path.join(dest, name)
        cl = per_cfg_line[name]
        head = name == legs[0][0]
        if head:
            workload, decoded = cl["config"]["workload"], cl["config"]["streams_total"] * cl["config"]["bytes_per_stream"]
            kernel_ms, algo = cl["roofline"]["kernel_ms"], cl["roofline"]["algorithmic_bytes_per_launch"]
            bl = {k: v for k, v in cl.items() if k != "configs"}
        else:
            workload, decoded = cl["workload"], cl["streams"] * cl["bytes_per_stream"]
            kernel_ms, algo = cl["kernel_ms"], cl["roofline"]["algorithmic_bytes_per_launch"]
            bl = cl
        json.dump(bl, open(pre + "_benchline.json", "w"))
        with open(pre + "_kernel_trace.csv", "w") as g:
            w = csv.DictWriter(g, fieldnames=header)
            w.writeheader()
            w.writerows(tr_by[name])
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr_by[name]]
        if head and head_stats:
            shutil.copy(head_stats[0], pre + "_kernel_stats.csv")
            stats_note = "rocprofv3's own --stats table of the one-config pass (`--configs none`)"
        else:
            with open(pre + "_kernel_stats.csv", "w") as g:
                w = csv.writer(g)
                w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
                w.writerow([tr_by[name][0]["Kernel_Name"], len(dur), sum(dur), sum(dur) / len(dur), min(dur), max(dur)])
            stats_note = "computed from this config's launches in the kernel trace (the --stats table of the whole process: all_kernel_stats.csv)"
        with open(pre + "_pmc.csv", "w") as g:
            w = csv.writer(g)
            w.writerow(SP.PMC_HEADER)
            w.writerows(rows_by[name])
        entry = SP.make_entry(workload, rev, decoded, kernel_ms, grid_by[name], vals_by[name], pre, wide_reads=name in WIDE_READS)
        SP.register(name, entry)
        vals = vals_by[name]
        with open(pre + "_summary.md", "w") as g:
            p = lambda *a: print(*a, file=g)
            p("# rocprofv3 summary (%s, prof_%s)\n" % (name, tag))
            p("command (one process per pass, every config in it; this file: the launches of %s): `python3 bench.py %s`\n"
              % (name, command.replace("bench args: ", "")))
            p("kernel id %s\n" % rev)
            for pn in ("kt", "fetch", "write", "sq", "sq2"):
                j = lines[pn]
                c = j if head else next(x for x in j["configs"] if x["name"] == name)
                p("- %s pass bench line: value %.3f GiB/s, kernel_ms %.3f, achieved %.3f GB/s"
                  % (pn, c["value"], c["roofline"]["kernel_ms"], c["roofline"]["achieved"]))
            p("\n## --kernel-trace (%s)\n" % stats_note)
            p("| kernel | calls | total ns | avg ns |\n|---|---|---|---|")
            p("| %s | %d | %d | %.0f |" % (tr_by[name][0]["Kernel_Name"], len(dur), sum(dur), sum(dur) / len(dur)))
            p("\nper launch of the decode kernel, in launch order (ms): %s -- the first is bench.py's untimed warm-up, the rest are "
              "the timed steps and agree with the HIP-event figure above." % ", ".join("%.3f" % (x / 1e6) for x in dur))
            p("\n## PMC (per launch of the decode kernel, summed over the device)\n")
            p("| counter | per launch (mean) | launches |\n|---|---|---|")
            for k, v in sorted(vals.items()):
                p("| %s | %.6g | %d |" % (k, sum(v) / len(v), len(v)))
            f = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024
            wr = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
            if name in WIDE_READS:
                p("\nHBM traffic per launch: FETCH_SIZE %.3f GB raw (KiB units x1024) = %.3f GB after the gfx950 correction (x 2: this "
                  "launch is the stored-chunk copy, 16-byte-per-lane streaming reads; the corrected figure equals the compressed bytes "
                  "the launch reads once), WRITE_SIZE %.3f GB; algorithmic bytes per launch %.3f GB."
                  % (f / 1e9, 2 * f / 1e9, wr / 1e9, algo / 1e9))
            else:
                p("\nHBM traffic per launch: FETCH_SIZE %.3f GB (raw, KiB units x1024; gfx950 reports half of wide streaming reads -- "
                  "this config's reads are byte / dword gathers, uncalibrated), WRITE_SIZE %.3f GB; algorithmic bytes per launch "
                  "%.3f GB." % (f / 1e9, wr / 1e9, algo / 1e9))
        print("%-10s kernel_ms %.3f  fetch %.2f GB  write %.2f GB  algorithmic %.2f GB" % (name, kernel_ms, f / 1e9, wr / 1e9, algo / 1e9))


if __name__ == "__main__":
    main()
