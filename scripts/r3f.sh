#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3f
MIA_NT_BK128=1 timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_nt" --timeout 120 --timeout-method thread > gpurun_out/r3f/tests.log 2>&1 || { tail -40 gpurun_out/r3f/tests.log; exit 1; }
tail -2 gpurun_out/r3f/tests.log
MIA_NT_BK128=0 timeout -k 10 300 python scripts/gemm_time.py 20 > gpurun_out/r3f/gt_bk64.txt 2>&1 || exit 1
MIA_NT_BK128=1 timeout -k 10 300 python scripts/gemm_time.py 20 > gpurun_out/r3f/gt_bk128.txt 2>&1 || exit 1
paste <(grep "us" gpurun_out/r3f/gt_bk64.txt | awk '{print $1,$2,$3,$4,$5,$6,$7}') <(grep "us" gpurun_out/r3f/gt_bk128.txt | awk '{print $7}') | awk '{ if ($7 != $8) print $0 }'
tail -1 gpurun_out/r3f/gt_bk64.txt; tail -1 gpurun_out/r3f/gt_bk128.txt
