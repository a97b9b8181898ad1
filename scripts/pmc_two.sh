#!/bin/bash
# usage: pmc_two.sh <tag> <kbench cases> : SQ wait/busy counters for kbench cases
set -e
TAG=$1; CASES=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_$TAG -o out --output-format csv -- python3 $R/scripts/kbench.py $CASES 3 > $R/gpurun_out/pmc_$TAG.log 2>&1
echo done
