#!/bin/bash
# resident waves per CU beyond 16 (needs more streams than 16 x 256 to matter)
cd "$(dirname "$0")/.."
for pc in 16 18 20; do
  echo "== per_cu $pc T 20480x64K"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 20480 65536 64 6 | grep "run 2"
  echo "== per_cu $pc T 5120x256K"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 5120 262144 64 6 | grep "run 2"
done
