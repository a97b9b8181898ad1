#!/bin/bash
# usage: pmc_three.sh <tag> <kbench cases> : matrix-core / LDS counters for kbench cases
set -e
TAG=$1; CASES=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/pmc_$TAG -o out --output-format csv -- python3 $R/scripts/kbench.py $CASES 3 > $R/gpurun_out/pmc_$TAG.log 2>&1
echo done
