#!/bin/bash
# Dev tool: build libxlz.so with a variant of the generated fast loop for an A/B run (tools/ab_bench.py).
#   tools/ab_build.sh <name> [--variant a,b] [--without c,d] [-- extra hipcc flags]
# writes lzma_amd/csrc/xlz_fastpath_<name>.inc (git-ignored) and build_ab/<name>.so (git-ignored, travels with gpurun).
set -eu
cd "$(dirname "$0")/.."
NAME=$1; shift
GEN=(); FLAGS=()
while [ $# -gt 0 ]; do
    case "$1" in
        --variant|--without) GEN+=("$1" "$2"); shift 2;;
        --) shift; FLAGS=("$@"); break;;
        *) echo "unknown argument $1"; exit 2;;
    esac
done
mkdir -p build_ab
python3 tools/gen_fastpath.py "${GEN[@]}" --out lzma_amd/csrc/xlz_fastpath_$NAME.inc | tail -1
# ... and the same variant over the compact model layout (launches whose units all have pb <= 2 run THAT loop)
GENC=(); HAVE_VARIANT=0
for ((i = 0; i < ${#GEN[@]}; i += 2)); do
    if [ "${GEN[i]}" = "--variant" ]; then GENC+=(--variant "${GEN[i+1]},compact"); HAVE_VARIANT=1; else GENC+=("${GEN[i]}" "${GEN[i+1]}"); fi
done
[ $HAVE_VARIANT = 1 ] || GENC+=(--variant compact)
python3 tools/gen_fastpath.py "${GENC[@]}" --out lzma_amd/csrc/xlz_fastpath_${NAME}_pb2.inc | tail -1
python3 - "$NAME" "${FLAGS[@]}" <<'PY'
import sys
sys.path.insert(0, ".")
from lzma_amd import build
name, flags = sys.argv[1], sys.argv[2:]
print(build.build(force=True, extra_flags=['-DXLZ_FASTPATH_INC="xlz_fastpath_%s.inc"' % name, '-DXLZ_FASTPATH_PB2_INC="xlz_fastpath_%s_pb2.inc"' % name] + flags,
                  out="build_ab/%s.so" % name))
PY
