"""Dev tool (GPU box): a few decode launches of one corpus family for rocprofv3's PC sampling, and the aggregation of
its samples by code offset (the raw CSV is too large to keep).
    rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles \\
              --pc-sampling-interval 4194304 -d <dir> --output-format csv -- python3 tools/pc_sample_run.py run T
    python3 tools/pc_sample_run.py aggregate <dir> > hist.json
The histogram is joined with the kernel's disassembly (llvm-objdump of the code object) by tools/pc_sample_report.py."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(fam):
    import corpus, lzma_amd
    nd, rep, size = 256, 16, 1 << 20
    cs, hs = corpus.make_alone_batch(fam, nd, size, base_seed=77, workers=min(os.cpu_count() or 1, 32), preset=6 if fam == "T" else 0)
    ctx = lzma_amd.Context(0)
    b = lzma_amd.Batch(ctx, [lzma_amd.Stream(cs[i % nd], out_cap=size) for i in range(nd * rep)])
    for _ in range(3):
        b.run()
    b.sync()
    res = b.results()
    assert all(r[1] == 0 and r[0] == size for r in res)
    print("ran 3 launches of %d streams, kernel %.1f ms" % (nd * rep, b.kernel_ms()))


def aggregate(d):
    files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
    out = {"files": [os.path.basename(f) for f in files]}
    for f in files:
        with open(f) as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames
            out.setdefault("columns", {})[os.path.basename(f)] = cols
            off = next((c for c in cols if "offset" in c.lower()), None)
            keyc = [c for c in cols if any(k in c.lower() for k in ("stall", "inst_type", "instruction_type", "issued", "reason", "arb"))]
            hist = {}
            n = 0
            for r in rd:
                n += 1
                k = (r.get("Code_Object_Id") or r.get("code_object_id") or "", r.get(off, "") if off else "")
                e = hist.setdefault("|".join(k), {"n": 0})
                e["n"] += 1
                for c in keyc:
                    kk = c + "=" + str(r[c])
                    e[kk] = e.get(kk, 0) + 1
            out.setdefault("samples", {})[os.path.basename(f)] = n
            out.setdefault("hist", {})[os.path.basename(f)] = hist
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "T")
    else:
        aggregate(sys.argv[2])
