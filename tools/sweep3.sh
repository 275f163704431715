#!/bin/bash
cd "$(dirname "$0")/.."
for pc in 4 6 7 8 9 10; do
  echo "== per_cu $pc R"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py R 5120 65536 64 1 | grep "run 2"
done
for pc in 4 6 7 8 9 10; do
  echo "== per_cu $pc T"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 5120 131072 64 6 | grep "run 2"
done
