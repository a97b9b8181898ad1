#!/bin/bash
# PMC pass over the GEMM micro-benchmarks (one counter set per run; kernel-trace only).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in gemm_exp10 gemm_proj11 gemm_exp6 gemm_proj6 tn_exp11 tn_proj6; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmcg/$c -o out --output-format csv -- python3 $R/scripts/kbench.py $c 3 > $R/gpurun_out/pmcg_$c.log 2>&1
done
echo done
