#!/bin/bash
# BASELINE.json configs 2-5 (+ the incompressible family) through bench.py on one GPU.
cd "$(dirname "$0")/.."; O=gpurun_out/configs_$1; mkdir -p $O
run() { name=$1; shift; echo "== $name: $@"; timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo FAILED; tail -5 $O/$name.err; }; tail -2 $O/$name.err; cat $O/$name.json; }
run cfg2_T --steps 5 --warmup 1
run cfg2_R --steps 3 --warmup 1 --family R --no-cpu-baseline --distinct 512
run cfg3 --steps 3 --warmup 1 --streams 65536 --size 65536 --no-cpu-baseline --verify sample
run cfg4_lzma2 --steps 3 --warmup 1 --format lzma2 --streams 1 --segments 4096 --size 262144 --no-cpu-baseline
run cfg5 --steps 2 --warmup 1 --streams 8192 --size 2097152 --lc 2 --lp 1 --pb 1 --dict 8388608 --distinct 512 --no-cpu-baseline --verify sample
