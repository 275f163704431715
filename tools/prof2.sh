#!/bin/bash
# latency-side counters of the decode kernel (separate PMC passes, never combined with tracing)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof2_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAM=${2:-T}
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p1 --output-format csv -- python3 $R/tools/prof_run.py $FAM 4096 262144 64 6 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CYCLES -d $O/p2 --output-format csv -- python3 $R/tools/prof_run.py $FAM 4096 262144 64 6 > $O/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in glob.glob("$O/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "xlz_decode" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(v): print("%-28s %.4g (n=%d)" % (k, sum(v[k]) / len(v[k]), len(v[k])))
PY
