#!/bin/bash
# Dev tool (GPU box): cfg4-R (one LZMA2 stream of stored chunks: the copy path) on several builds of the library.
#   tools/ab_stored.sh name=path.so [name=path.so ...]      (the tree's own library runs first as "tree")
set -u
R=${GRAFT_REPO_ROOT:-.}
ARGS="--headline cfg4-R --configs none --extras none --no-cpu-baseline --steps 20 --warmup 3 --corpus-cache /tmp/xlz_corpus_cache"
one() { python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %9.1f GiB/s  kernel %.4f ms  frac %.4f  occupancy %s' % ('$1', l['value'], l['roofline']['kernel_ms'], l['roofline']['frac'], l['roofline'].get('issue',{}).get('slot_occupancy')))"; }
python3 $R/bench.py $ARGS 2>/dev/null | one tree
for a in "$@"; do
    XLZ_SO=$R/${a#*=} python3 $R/bench.py $ARGS --allow-xlz-so 2>/dev/null | one ${a%%=*}
done
