#!/bin/bash
# Issue-slot counters of single kernels (GPU box only):  bash scripts/pmc_kbench.sh TAG case1,case2,...
# Two counter passes (the SQ block holds eight counters a pass), each a run of its own under a hard time limit;
# per-kernel means are printed by scripts/pmc_show.py.
TAG=$1; CASES=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
i=1
for P in "$P1" "$P2"; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $P -d $R/gpurun_out/$TAG/p$i -o k --output-format csv -- python3 $R/scripts/kbench.py $CASES 3 > $R/gpurun_out/$TAG/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/$TAG/p$i.log; exit 1; }
  i=$((i+1))
done
for i in 1 2; do python3 $R/scripts/pmc_show.py $R/gpurun_out/$TAG/p$i dwconv; python3 $R/scripts/pmc_show.py $R/gpurun_out/$TAG/p$i gemm; done
