#!/bin/bash
cd "$(dirname "$0")/.."
for pc in 4 8 10; do XLZ_PER_CU=$pc timeout -k 10 100 python tools/exp_waves.py 10240; done
for pc in 8 10 12 14 16 20; do XLZ_SO=$PWD/lzma_amd/libxlz_exp.so XLZ_PER_CU=$pc timeout -k 10 100 python tools/exp_waves.py 10240; done
