#!/bin/bash
# parity suite, then residency sweep on the T and R families (XLZ_PER_CU caps resident waves per CU)
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 || exit 1
for pc in ${PCS:-16}; do
  echo "== per_cu $pc T 4096x256K"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 4096 262144 64 6 | grep "run 2"
  echo "== per_cu $pc T 16384x64K"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 16384 65536 64 6 | grep "run 2"
  echo "== per_cu $pc R 16384x16K"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py R 16384 16384 64 1 | grep "run 2"
done
