#!/bin/bash
# families M and Z at the cfg2 shape, and the cfg5 variant with distances up to 8 MiB and window wrap
cd "$(dirname "$0")/.."; O=gpurun_out/extra_$1; mkdir -p $O
run() { name=$1; shift; echo "== $name: $@"; timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo FAILED; tail -5 $O/$name.err; }; tail -1 $O/$name.err; cat $O/$name.json; }
run cfg2_M --steps 3 --warmup 1 --family M --no-cpu-baseline --distinct 512
run cfg2_Z --steps 3 --warmup 1 --family Z --no-cpu-baseline --distinct 512
run cfg5_wrap --steps 2 --warmup 1 --streams 64 --size 25165824 --lc 2 --lp 1 --pb 1 --dict 8388608 --distinct 16 --no-cpu-baseline
