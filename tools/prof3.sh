#!/bin/bash
# instruction-cache side of the decode kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof3_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU -d $O/p1 --output-format csv -- python3 $R/tools/prof_run.py T 4096 262144 64 6 > $O/p1.log 2>&1
tail -3 $O/p1.log
python3 - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$O/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "xlz_decode" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(v): print("%-28s %.4g (n=%d)" % (k, sum(v[k]) / len(v[k]), len(v[k])))
PY
