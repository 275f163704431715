timeout -k 10 300 python tools/ab_bench.py base=build_ab/base.so smask=build_ab/smask.so --fams T,R,S 2>&1 | tee gpurun_out/ab7.log
cd /tmp && export TMPDIR=/tmp
for v in base smask; do
XLZ_SO=$GRAFT_REPO_ROOT/build_ab/$v.so rocprofv3 --pmc WRITE_SIZE FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_$v/wr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --headline cfg2-T --configs none --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${v}_wr.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_${v}_wr.err; echo $v rc $?
done
python3 - <<'PY'
import csv,glob,os
R=os.environ["GRAFT_REPO_ROOT"]
for tag in ("prof_base/wr","prof_smask/wr"):
  for f in glob.glob(R+"/gpurun_out/"+tag+"/*/*counter_collection.csv"):
    for cn in ("WRITE_SIZE","FETCH_SIZE"):
        v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "xlz_decode" in r["Kernel_Name"] and r["Counter_Name"]==cn]
        print(tag, cn, "per launch GB:", [round(x*1024/1e9,2) for x in v])
PY
