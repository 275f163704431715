#!/bin/bash
# rocprofv3 passes over EVERY bench.py config, one process per pass (run on the GPU box via gpurun): the corpora are
# generated once (first pass, kept in /tmp) and every pass decodes cfg3 and the side configs in bench.py's order, so a
# pass costs one python start instead of one per config (tools/profile_bench.sh: one config, six processes).
#   tools/profile_all.sh <tag> [comma list of side configs]
# 1) --kernel-trace --stats  2) --pmc FETCH_SIZE  3) --pmc WRITE_SIZE  4) SQ instruction mix  5) SQ issue activity
# Counter passes are separate runs and never combined with tracing (pool rule).  The program after `--` is python3
# itself (no env / shell hop: the profiler has initialised the GPU by then).  tools/save_profiles_all.py cuts the
# dispatch lists into configs (every leg is warm-up + steps launches, in the order of the command line).
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun (GRAFT_REPO_ROOT is set there)}"
TAG=${1:-r04all}
SIDE=${2:-cfg2-T,cfg2-R,cfg4,cfg4-R,cfg5,cfg5-wrap}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--headline cfg3 --configs $SIDE --extras none --steps 3 --warmup 1 --side-steps 3 --no-cpu-baseline --corpus-cache /tmp/xlz_corpus_cache"
echo "bench args: $ARGS" > $O/command.txt
run() { # name, rocprofv3 options
    local n=$1; shift
    echo "[profile_all] pass $n: $(date +%T)"
    rocprofv3 "$@" -d $O/$n --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/$n.json > $O/$n.line 2> $O/$n.err || { tail -5 $O/$n.err; exit 1; }
}
run kt --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
# the headline alone in its process: rocprofv3's own --stats table then holds nothing but the headline's launches
ARGS="--headline cfg3 --configs none --extras none --steps 3 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/xlz_corpus_cache"
run kt_head --kernel-trace --stats
echo "[profile_all] done: $(date +%T)"
# keep what tools/save_profiles_all.py needs small enough for gpurun's 64 MiB return
find $O -name '*agent_info.csv' -delete
