#!/bin/bash
# dev sweep: residency and LDS write mode
cd "$(dirname "$0")/.."
for pc in 1 2 4 6 8 10; do
  echo "== per_cu $pc (w0) R"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py R 2560 65536 64 1 | grep "run 2"
done
for pc in 1 4 10; do
  echo "== per_cu $pc (w1) R"; XLZ_SO=$PWD/lzma_amd/libxlz_w1.so XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py R 2560 65536 64 1 | grep "run 2"
done
for pc in 1 4 10; do
  echo "== per_cu $pc (w0) T"; XLZ_PER_CU=$pc timeout -k 10 200 python tools/gpu_quick.py T 2560 262144 64 6 | grep "run 2"
done
echo "== per_cu 10 (w1) T"; XLZ_SO=$PWD/lzma_amd/libxlz_w1.so timeout -k 10 200 python tools/gpu_quick.py T 2560 262144 64 6 | grep "run 2"
