#!/bin/bash
# default bench workload (4096 x 1 MiB text) against resident waves per CU
cd "$(dirname "$0")/.."
for pc in ${PCS:-16 20}; do
  echo "== per_cu $pc"; XLZ_PER_CU=$pc timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 ${BENCH_ARGS} | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['unit'], j['ms_per_step'], 'ms', j['roofline']['frac'])"
done
