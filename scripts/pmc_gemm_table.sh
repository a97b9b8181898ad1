#!/bin/bash
# Per-shape GEMM table with matrix-core counters: one kernel-trace run, one --pmc run (kernel-trace only beside it).
# usage: bash scripts/pmc_gemm_table.sh TAG
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o g --output-format csv -- python3 $R/scripts/gemm_shapes.py 6 > $O/cases.log 2> $O/trace.err
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc -o g --output-format csv -- python3 $R/scripts/gemm_shapes.py 6 > $O/cases_pmc.log 2> $O/pmc.err
cd $R
python scripts/gemm_table.py $O/cases.log $(ls $O/trace/*kernel_trace.csv $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1) $(ls $O/pmc/*counter_collection.csv $O/pmc/*/*counter_collection.csv 2>/dev/null | head -1) 6 > $O/gemm_table.md
cat $O/gemm_table.md
