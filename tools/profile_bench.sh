#!/bin/bash
# rocprofv3 passes over ONE bench.py config (run on the GPU box via gpurun).
#   tools/profile_bench.sh <tag> <config> [extra bench args]     config: cfg3 (default), cfg2-T, cfg2-R, cfg4, cfg5, ...
# 1) --kernel-trace --stats  2) --pmc FETCH_SIZE  3) --pmc WRITE_SIZE  4) SQ instruction mix  5) SQ issue activity
# 6) VmemLatency / LdsLatency
# Counter passes are separate runs and never combined with tracing (pool rule).  The program after
# `--` is python3 itself (no env / shell hop: the profiler has initialised the GPU by then).
set -u
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun (GRAFT_REPO_ROOT is set there)}"
TAG=${1:-r03}; CFG=${2:-cfg3}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--headline $CFG --configs none --extras none --steps 3 --warmup 1 --no-cpu-baseline --corpus-cache /tmp/xlz_corpus_cache $@"
echo "bench args: $ARGS" > $O/command.txt
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/kt.json > $O/kt.line 2> $O/kt.err || exit 1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/fetch.json > $O/fetch.line 2> $O/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/write.json > $O/write.line 2> $O/write.err || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/sq.json > $O/sq.line 2> $O/sq.err || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq2 --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/sq2.json > $O/sq2.line 2> $O/sq2.err || exit 1
# 6) average latency of the vector-memory and LDS instructions (derived metrics; optional: a failure here does not fail the run)
rocprofv3 --pmc VmemLatency LdsLatency -d $O/lat --output-format csv -- python3 $R/bench.py $ARGS --detail-out $O/lat.json > $O/lat.line 2> $O/lat.err || echo "latency pass failed (see lat.err)"
rm -f /tmp/xlz_corpus_cache/xlz_corpus_${CFG}_*.pkl
python3 $R/tools/summarize_prof.py $O > $O/summary.md
cat $O/summary.md
