#!/bin/bash
# default bench workload (4096 x 1 MiB text) against resident waves per CU
cd "$(dirname "$0")/.."
for pc in 12 16 20; do
  echo "== per_cu $pc"; XLZ_PER_CU=$pc timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 | tail -1
done
