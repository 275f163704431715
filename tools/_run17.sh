python bench.py --steps 20 --warmup 5 > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err; echo bench rc $?; tail -3 gpurun_out/r2h_bench.err
python -c "
import json; j=json.load(open('gpurun_out/r2h_bench.json')); print(j['value'], j['roofline']['issue'].get('slot_occupancy'), [(c['name'], c['value'], c['cpu_baseline']['value']) for c in j['configs']])"
bash tools/profile_bench.sh r02c cfg2-T > gpurun_out/r2h_prof_T.log 2>&1; echo profT rc $?
bash tools/profile_bench.sh r02d cfg2-R > gpurun_out/r2h_prof_R.log 2>&1; echo profR rc $?
