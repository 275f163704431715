#!/bin/bash
# PMC passes over the decode kernel (separate passes; never combined with tracing)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAM=${2:-R}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM -d $O/p1 --output-format csv -- python3 $R/tools/prof_run.py $FAM 2560 65536 64 2 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $O/p2 --output-format csv -- python3 $R/tools/prof_run.py $FAM 2560 65536 64 2 > $O/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS GRBM_GUI_ACTIVE -d $O/p3 --output-format csv -- python3 $R/tools/prof_run.py $FAM 2560 65536 64 2 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/tools/prof_run.py $FAM 2560 65536 64 2 > $O/kt.log 2>&1
find $O -name "*.csv" | head -20
