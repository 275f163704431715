"""Copy the judged parts of a tools/profile_bench.sh run from gpurun_out/ into profiles/.
usage: python tools/save_profile.py <tag> <dest-prefix>   e.g.  r01b profiles/r01/bench_default_v3"""
import csv, glob, json, os, shutil, sys
tag, dest = sys.argv[1], sys.argv[2]
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
                rows.append([sub, r["Dispatch_Id"], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"],
                             r["Counter_Name"], r["Counter_Value"]])
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
w = csv.writer(open(dest + "_pmc.csv", "w"))
w.writerow(["pass", "dispatch", "grid", "wg", "vgpr", "sgpr", "counter", "value"])
w.writerows(rows)
line = json.loads(open(src + "/kt.json").read().strip().splitlines()[-1])
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
traffic = {
    "workload": line["config"]["workload"],
    "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
    "traffic_bytes_per_launch": fetch + write,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units x 1024), mean over the launches "
            "of xlz_decode_kernel; FETCH_SIZE taken raw: this kernel's reads are one-byte-per-lane gathers and 4-byte-per-"
            "lane window loads, not the 16-byte-per-lane streaming reads the gfx950 x2 correction is calibrated for",
    "source": dest + "_pmc.csv",
}
json.dump(traffic, open(os.path.join(os.path.dirname(dest), "..", "traffic_default.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
