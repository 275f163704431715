#!/bin/bash
# Dev tool (GPU box): cfg4-R (one LZMA2 stream of stored chunks) with runs of stored chunks cut into units of at least
# N KiB (XLZ_STORED_UNIT_KIB; 0 = at dictionary resets only).      tools/ab_stored_unit.sh 0 64 128 256
# Needs a library built with -DXLZ_DEV_KNOBS (the shipped build ignores the variable):
#   python3 -c "from lzma_amd import build; build.build(extra_flags=['-DXLZ_DEV_KNOBS'], out='/tmp/libxlz_knobs.so')"
#   XLZ_SO=/tmp/libxlz_knobs.so tools/ab_stored_unit.sh ...   (bench.py: --allow-xlz-so is passed below)
set -u
R=${GRAFT_REPO_ROOT:-.}
ARGS="--headline cfg4-R --configs none --extras none --no-cpu-baseline --steps 20 --warmup 3 --corpus-cache /tmp/xlz_corpus_cache --allow-xlz-so"
for k in "$@"; do
    XLZ_STORED_UNIT_KIB=$k python3 $R/bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=l['roofline']
print('unit >= %4s KiB  %9.1f GiB/s  kernel %.4f ms  frac %.4f  units %d  occupancy %s' % ('$k', l['value'], r['kernel_ms'], r['frac'], r['units_per_launch'], r.get('slot_occupancy')))"
done
