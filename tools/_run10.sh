timeout -k 10 300 python tools/ab_bench.py rank=build_ab/rank.so rank64=build_ab/rank64.so --fams T,R,S 2>&1 | tee gpurun_out/ab5.log
XLZ_SO=$GRAFT_REPO_ROOT/build_ab/rank.so python bench.py --steps 3 --warmup 1 --configs cfg2-R,cfg3,cfg4,cfg5 --no-cpu-baseline --trace-out gpurun_out/trace_rank_ > gpurun_out/r2e_bench_rank.json 2> gpurun_out/r2e_bench_rank.err; echo bench rc $?
python -c "
import json; j=json.load(open('gpurun_out/r2e_bench_rank.json')); print(j['value'], j['roofline']['issue']['slot_occupancy'], [(c['name'], c['value'], c['roofline']['issue']['slot_occupancy']) for c in j['configs']])"
