python -m pytest tests -m gpu -x -q > gpurun_out/r2f_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r2f_pytest.log
bash tools/profile_bench.sh r02b cfg2-T > gpurun_out/r2f_prof_T.log 2>&1; echo profT rc $?
cd /tmp && export TMPDIR=/tmp
XLZ_SO=$GRAFT_REPO_ROOT/build_ab/cmask.so rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_cmask/fetch --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --headline cfg2-T --configs none --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_cmask_fetch.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_cmask_fetch.err; echo cmask fetch rc $?
python3 - <<'PY'
import csv,glob,os
R=os.environ["GRAFT_REPO_ROOT"]
for tag in ("prof_cmask/fetch","prof_r02b/fetch"):
  for f in glob.glob(R+"/gpurun_out/"+tag+"/*/*counter_collection.csv"):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "xlz_decode" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
    print(tag, "FETCH_SIZE per launch GB:", [round(x*1024/1e9,2) for x in v])
PY
