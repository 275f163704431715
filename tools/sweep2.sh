#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for pc in 1 4 10; do
  echo "== per_cu $pc R"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py R 2560 65536 64 1 | grep "run 2"
done
for pc in 1 4 10; do
  echo "== per_cu $pc T"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 2560 262144 64 6 | grep "run 2"
done
