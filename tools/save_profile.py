"""Copy the judged parts of a tools/profile_bench.sh run from gpurun_out/ into profiles/ and
register it in profiles/current.json (what bench.py reports as roofline.traffic / roofline.issue
while the kernel sources are the ones the profile was taken on).
usage: python tools/save_profile.py <tag> <dest-prefix> [config]   e.g.  r02a profiles/r02/cfg2-T_v9 cfg2-T
(tools/save_profiles_all.py does the same for a tools/profile_all.sh run: every config from ONE process per pass)"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CUS, CLK = 256, 2.4e9
PMC_HEADER = ["pass", "dispatch", "grid", "wg", "vgpr", "sgpr", "counter", "value"]


def pmc_row(sub, r):
    return [sub, r["Dispatch_Id"], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Counter_Name"],
            r["Counter_Value"]]


def make_entry(workload, kernel_rev, decoded, kernel_ms, grid_threads, vals, dest, wide_reads=False):
    """the profiles/current.json entry of one config: vals = {counter: [per-launch values]}, decoded = bytes decoded per
    launch, kernel_ms = the bench line's HIP-event figure of the same run, grid_threads = the launch's grid size"""
    mean = lambda k: sum(vals[k]) / len(vals[k])
    # wide_reads: the launch reads 16 bytes per lane in streaming order (the stored-chunk copy): MI355X_MICROARCH.md's
    # gfx950 correction applies (FETCH_SIZE tallies such requests at half their size) -- and is confirmed by the known
    # byte count of that config (2 x FETCH_SIZE = the compressed bytes read once)
    fetch_raw = mean("FETCH_SIZE") * 1024
    fetch = fetch_raw * (2 if wide_reads else 1)
    write = mean("WRITE_SIZE") * 1024
    grid = grid_threads // 64  # single-wave workgroups = wave slots of the launch
    cu_cycles = kernel_ms / 1e3 * CLK * CUS
    return {
        "workload": workload, "kernel_rev": kernel_rev, "source": dest + "_pmc.csv",
        "fetch_bytes_per_launch_raw": fetch_raw, "fetch_correction": 2 if wide_reads else 1, "write_bytes_per_launch": write,
        "traffic_bytes_per_launch": fetch + write,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units x 1024), mean over the launches "
                "of xlz_decode_kernel; " + (
                    "FETCH_SIZE x 2: this launch reads 16 bytes per lane in streaming order (stored chunks), the access the "
                    "gfx950 correction of MI355X_MICROARCH.md is calibrated for -- 2 x FETCH_SIZE equals the compressed bytes"
                    if wide_reads else
                    "FETCH_SIZE taken raw: this kernel's reads are one-byte-per-lane gathers and 4-byte-per-"
                    "lane window loads, not the 16-byte-per-lane streaming reads the gfx950 x2 correction is calibrated for"),
        "issue": {
            "salu_per_cu_cycle": round(mean("SQ_INSTS_SALU") / cu_cycles, 4),
            "valu_per_cu_cycle": round(mean("SQ_INSTS_VALU") / cu_cycles, 4),
            "branch_per_cu_cycle": round(mean("SQ_INSTS_BRANCH") / cu_cycles, 4),
            "lds_per_cu_cycle": round(mean("SQ_INSTS_LDS") / cu_cycles, 4),
            # measured ceilings (tools/ubench/mix2.hip, 16 waves per CU): 0.97 SALU and 1.28 simple wave64 VALU
            # instructions per CU cycle, both reachable at once by a strictly alternating stream.  Branches are not
            # on the SALU port (SALU + branch exceeds 0.97 on incompressible data).
            "salu_utilisation": round(mean("SQ_INSTS_SALU") / cu_cycles / 0.97, 4),
            "valu_utilisation": round(mean("SQ_INSTS_VALU") / cu_cycles / 1.28, 4),
            "instructions_per_decoded_byte": round((mean("SQ_INSTS_SALU") + mean("SQ_INSTS_VALU") + mean("SQ_INSTS_BRANCH") +
                                                   mean("SQ_INSTS_LDS") + mean("SQ_INSTS_VMEM")) / decoded, 2),
            # what bench.py's roofline.issue.issue_bound is made of
            "salu_per_decoded_byte": round(mean("SQ_INSTS_SALU") / decoded, 3),
            "valu_per_decoded_byte": round(mean("SQ_INSTS_VALU") / decoded, 3),
            "branch_per_decoded_byte": round(mean("SQ_INSTS_BRANCH") / decoded, 3),
            # SQ_WAVE_CYCLES counts in quad-cycles per wave: x4 / (slots x kernel cycles) = average slot occupancy
            "slot_occupancy": round(mean("SQ_WAVE_CYCLES") * 4 / (grid * kernel_ms / 1e3 * CLK), 4),
            # where a resident wave's time goes: the three counters add up to SQ_WAVE_CYCLES
            "wave_time": {"executing_an_instruction": round(mean("SQ_ACTIVE_INST_ANY") / mean("SQ_WAVE_CYCLES"), 3),
                          "in_s_waitcnt": round(mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES"), 3),
                          "waiting_to_issue": round(mean("SQ_WAIT_INST_ANY") / mean("SQ_WAVE_CYCLES"), 3)},
            "budget": "measured ceilings per CU cycle: 0.97 SALU, 1.28 VALU (tools/ubench/mix2.hip); "
                      "clock %.1f GHz, %d CUs, kernel_ms %.3f" % (CLK / 1e9, CUS, kernel_ms),
            "source": dest + "_pmc.csv",
        },
    }


def register(cfg, entry):
    cur_path = os.path.join(ROOT, "profiles", "current.json")
    cur = json.load(open(cur_path)) if os.path.exists(cur_path) else {}
    cur[cfg] = entry
    json.dump(cur, open(cur_path, "w"), indent=1)


def main():
    import bench  # kernel_rev(): the build id compiled into lzma_amd/libxlz.so (must be the library the profile ran on)
    tag, dest = sys.argv[1], sys.argv[2]
    cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2-T"
    src = "gpurun_out/prof_" + tag
    os.makedirs(os.path.dirname(dest), exist_ok=True)
    shutil.copy(src + "/summary.md", dest + "_summary.md")
    shutil.copy(glob.glob(src + "/kt/*/*kernel_stats.csv")[0], dest + "_kernel_stats.csv")
    shutil.copy(src + "/kt.json", dest + "_benchline.json")
    with open(glob.glob(src + "/kt/*/*kernel_trace.csv")[0]) as f, open(dest + "_kernel_trace.csv", "w") as g:
        for i, l in enumerate(f):
            if i == 0 or "xlz_decode" in l:
                g.write(l)
    rows = []
    vals = {}
    for sub in ("fetch", "write", "sq", "sq2"):
        for f in glob.glob(src + "/%s/*/*counter_collection.csv" % sub):
            for r in csv.DictReader(open(f)):
                if "xlz_decode" in r["Kernel_Name"]:
                    rows.append(pmc_row(sub, r))
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    w = csv.writer(open(dest + "_pmc.csv", "w"))
    w.writerow(PMC_HEADER)
    w.writerows(rows)
    line = json.load(open(src + "/kt.json"))   # bench.py --detail-out: the full record
    assert line["config"]["kernel_rev"] == bench.kernel_rev(), "the profile was taken on another build than lzma_amd/libxlz.so"
    decoded = line["config"]["streams_total"] * line["config"]["bytes_per_stream"]
    entry = make_entry(line["config"]["workload"], line["config"]["kernel_rev"], decoded, line["roofline"]["kernel_ms"],
                       int(rows[0][2]), vals, dest)
    register(cfg, entry)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
