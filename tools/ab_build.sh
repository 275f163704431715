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
python3 - "$NAME" "${FLAGS[@]}" <<'PY'
import sys
sys.path.insert(0, ".")
from lzma_amd import build
name, flags = sys.argv[1], sys.argv[2:]
print(build.build(force=True, extra_flags=['-DXLZ_FASTPATH_INC="xlz_fastpath_%s.inc"' % name] + flags, out="build_ab/%s.so" % name))
PY
